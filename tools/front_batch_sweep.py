#!/usr/bin/env python3
"""Front-pass size of the static CNN against time (round-5 review item 1: can the front's trunk stay resident in the 256 MiB
memory-side cache?).  The front (stem, stage 1, first block of stage 2) runs in passes of F frames, the back once over all
2048; results are bit-identical for every F (tests/test_gpu_edges.py).  Per F: wall time of the 2048-frame call without events,
then the per-family HIP-event sums of a second set of calls.  Judged by TIME (FETCH_SIZE counts Infinity-Cache hits as traffic).

    python tools/front_batch_sweep.py [frames=2048] [F ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from avcer_amd import synth  # noqa: E402
from avcer_amd.engine import MODE_F16X3, Engine  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    fronts = [int(a) for a in sys.argv[2:]] or [1024, 512, 256, 128, 96, 64, 48, 40, 32, 30, 24, 20, 16, 12, 8, 1024]
    eng = Engine(0)
    eng.load_static(synth.static_state_dict(42))
    frames = torch.from_numpy(synth.face_frames(1234, n)).cuda()
    iters = 4
    print(f"{n} frames per call, x3 mode, back pass {min(n, 2048)}; ms per call (median of {iters}); families from a second set of calls with events")
    print(f"{'front':>6} {'wall':>8} {'events':>8} | {'chains':>8} {'wd':>8} {'staged':>8} {'tail':>8} {'stem':>8} {'skinny':>8} | trunk MB (OUT + T1' of one pass)")
    base = None
    for f in fronts:
        eng.set_static_batch(f, back=min(n, 2048))
        for _ in range(2):
            eng.static_forward(frames, MODE_F16X3)
        torch.cuda.synchronize()
        dts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            eng.static_forward(frames, MODE_F16X3)
            torch.cuda.synchronize()
            dts.append((time.perf_counter() - t0) * 1e3)
        wall = sorted(dts)[len(dts) // 2]
        eng.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(iters):
            eng.static_forward(frames, MODE_F16X3)
        torch.cuda.synchronize()
        ev_wall = (time.perf_counter() - t0) * 1e3 / iters
        fam = eng.profile_read_families()
        eng.profile_enable(False)
        g = lambda k: fam[k][0] / iters  # noqa: E731
        trunk = f * 55 * 55 * (256 + 64) * 4 / 1e6
        base = base or wall
        print(f"{f:6d} {wall:8.2f} {ev_wall:8.2f} | {g('bneck_kernel'):8.2f} {g('conv_gemm_wd_kernel'):8.2f} {g('conv_gemm_kernel'):8.2f} "
              f"{g('bneck_tail2_kernel'):8.2f} {g('stem_pool_kernel'):8.2f} {g('conv_gemm_skinny_kernel'):8.2f} | {trunk:7.1f}  ({wall / base - 1:+.1%})",
              flush=True)
    eng.set_static_batch(1024, back=0)
