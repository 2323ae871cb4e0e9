#!/usr/bin/env python3
"""Where does the stage-3 tail (conv3 + residual + next conv1) as TWO contractions beat the fused bneck_tail2_kernel?  (lab)
Static CNN, x3 mode, frames per call, always fused against always the pair, both timed in one process (one context each).
The knob it drives (AVCER_LAB_TAIL_PAIR_ROWS, read at context creation) existed in a lab build only: the product has the
measured threshold as a constant (api.hip kTailPairRows).  Result: profiles/experiments/r05_tail_pair_probe.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from avcer_amd import synth  # noqa: E402
from avcer_amd.engine import MODE_F16X3, Engine  # noqa: E402

sd = synth.static_state_dict(42)
engs = {}
for rows in (0, 1 << 40):
    os.environ["AVCER_LAB_TAIL_PAIR_ROWS"] = str(rows)
    engs[rows] = Engine(0)
    engs[rows].load_static(sd)
for n in [int(a) for a in sys.argv[1:]] or [1, 4, 16, 64, 128, 256, 512, 2048]:
    frames = torch.from_numpy(synth.face_frames(1, n)).cuda()
    out = {}
    for rows, e in engs.items():
        for _ in range(3):
            r = e.static_forward(frames, MODE_F16X3)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(10 if n <= 256 else 3):
                e.static_forward(frames, MODE_F16X3)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / (10 if n <= 256 else 3) * 1e3)
        out[rows] = (sorted(ts)[2], r)
    same = all(torch.equal(a, b) for a, b in zip(out[0][1], out[1 << 40][1]))
    print(f"frames {n:5d} (tail rows {n * 196:7d})  fused {out[0][0]:8.3f} ms   pair {out[1 << 40][0]:8.3f} ms   bit-identical {same}", flush=True)
