#!/usr/bin/env python3
"""BASELINE.json configs 2 and 3 measured on their own (GPU): static CNN at batch 256 / 1024 and the audio model at
128 windows of 2 s, in every arithmetic mode.  Prints frames/s, windows/s and the MFMA fractions
(algorithmic FLOPs: 6.927 GFLOP/frame -- the reference graph's 7.667 less the 0.74 nothing reads --, 44.891 GFLOP/window; x3 executes 3 MFMA products per algorithmic product)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avcer_amd import synth  # noqa: E402
from avcer_amd.engine import MODE_BF16, MODE_F16X3, MODE_FP32, Engine  # noqa: E402

MODES = (("fp32", MODE_FP32, 157.3, 1), ("bf16", MODE_BF16, 2500.0, 1), ("x3", MODE_F16X3, 2500.0, 3))


def timeit(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


if __name__ == "__main__":
    eng = Engine(0)
    eng.load_static(synth.static_state_dict(42))
    eng.load_audio(synth.audio_state_dict(42))
    for batch in (256, 1024):
        frames = torch.from_numpy(synth.face_frames(1, batch)).to(eng.device)
        for name, mode, peak, passes in MODES:
            dt = timeit(lambda: eng.static_forward(frames, mode))
            tf = 6.927e9 * batch / dt / 1e12  # products whose results are read (bench.py GFLOP_STATIC_FRAME)
            print(f"static CNN  batch {batch:5d} {name:5s}: {dt*1e3:8.2f} ms  {batch/dt:9.0f} frames/s  {tf:7.1f} TFLOP/s algorithmic "
                  f"= {tf/peak:.3f} of {peak:.0f}  (executed MFMA fraction {tf*passes/peak:.3f})")
    wav = torch.from_numpy(synth.waveforms(2, 128, 32000)).to(eng.device)
    for name, mode, peak, passes in MODES:
        dt = timeit(lambda: eng.audio_forward(wav, True, mode))
        tf = 44.891e9 * 128 / dt / 1e12
        print(f"audio model 128 x 2 s     {name:5s}: {dt*1e3:8.2f} ms  {128/dt:9.0f} windows/s {tf:7.1f} TFLOP/s algorithmic "
              f"= {tf/peak:.3f} of {peak:.0f}  (executed MFMA fraction {tf*passes/peak:.3f})")
