#!/usr/bin/env python3
"""Which call sizes of avcer_static_forward gain from the two-lane schedule (two half-batches on two HIP streams)?
ms per call, one lane against two lanes, alternating, median of 7 rounds of 10 calls.   python tools/two_lane_sweep.py [n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from avcer_amd import synth  # noqa: E402
from avcer_amd.engine import MODE_F16X3, Engine  # noqa: E402

if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [64, 96, 128, 192, 256, 384, 512, 640, 750, 896, 1024]
    eng = Engine(0)
    eng.load_static(synth.static_state_dict(42))
    frames_all = torch.from_numpy(synth.face_frames(1, max(sizes))).cuda()
    print(f"{'frames':>7s} {'one lane ms':>12s} {'two lanes ms':>13s} {'change':>8s}")
    for n in sizes:
        frames = frames_all[:n]
        res = {1: [], 2: []}
        for lanes in (1, 2):
            eng.set_static_lanes(lanes, 2, 4096)
            for _ in range(3):
                eng.static_forward(frames, MODE_F16X3)
        torch.cuda.synchronize()
        for _ in range(7):
            for lanes in (1, 2):
                eng.set_static_lanes(lanes, 2, 4096)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    eng.static_forward(frames, MODE_F16X3)
                torch.cuda.synchronize()
                res[lanes].append((time.perf_counter() - t0) / 10 * 1e3)
        a, b = sorted(res[1])[3], sorted(res[2])[3]
        print(f"{n:7d} {a:12.3f} {b:13.3f} {b / a - 1:+8.1%}", flush=True)
