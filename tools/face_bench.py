#!/usr/bin/env python3
"""Throughput of the RetinaFace-R50 detector network (row f4) on synthetic video frames, per arithmetic mode."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avcer_amd import synth  # noqa: E402
from avcer_amd.engine import Engine, MODE_BF16, MODE_F16X3, MODE_FP32  # noqa: E402

if __name__ == "__main__":
    eng = Engine(0)
    eng.load_face(synth.to_torch(synth.retina_state_dict(42)))
    for h, w, n in ((360, 640, 32), (720, 1280, 8)):
        frames = torch.from_numpy(synth.video_frames(3, 2, h, w)).cuda().repeat(n // 2, 1, 1, 1)
        for name, mode in (("fp32", MODE_FP32), ("x3", MODE_F16X3), ("bf16", MODE_BF16)):
            for _ in range(2):
                eng.face_forward(frames, mode)
            torch.cuda.synchronize()
            eng.gemm_stats(reset=True)
            t0 = time.perf_counter()
            for _ in range(3):
                eng.face_forward(frames, mode)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            launches, flops = eng.gemm_stats(reset=True)
            print(f"{h}x{w} batch {n:3d} {name:5s}: {dt * 1e3:8.2f} ms  {n / dt:8.1f} frames/s  "
                  f"{flops / 3 / dt / 1e12:6.1f} TFLOP/s algorithmic ({flops / 3 / n / 1e9:.1f} GFLOP/frame)")
