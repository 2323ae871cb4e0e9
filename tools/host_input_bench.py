#!/usr/bin/env python3
"""PCIe-inclusive rate of the bench step: the same 128-clip step as bench.py, but with the u8 tiles and waveforms
starting in pinned HOST memory every step (copied on a side stream, double-buffered, overlapped with the previous
step's kernels).  Reported in DESIGN.md next to the HBM-resident headline; never the bench `value`."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from avcer_amd import synth  # noqa: E402
from avcer_amd.engine import MODE_F16X3  # noqa: E402
from avcer_amd.pipeline import AVPipeline  # noqa: E402

if __name__ == "__main__":
    clips, steps, warmup = 128, 5, 2
    pipe = AVPipeline(0, seed=42, mode=MODE_F16X3)
    dev = pipe.engine.device
    h_frames = torch.from_numpy(synth.face_frames(1234, clips * bench.T_FRAMES).reshape(clips, bench.T_FRAMES, 224, 224, 3)).pin_memory()
    h_wav = torch.from_numpy(synth.waveforms(5678, clips, bench.T_AUDIO)).pin_memory()
    copy = torch.cuda.Stream()
    bufs = [(torch.empty_like(h_frames, device=dev), torch.empty_like(h_wav, device=dev)) for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]

    def upload(i):
        with torch.cuda.stream(copy):
            copy.wait_event(done[i])           # the step that last read this buffer has finished
            bufs[i][0].copy_(h_frames, non_blocking=True)
            bufs[i][1].copy_(h_wav, non_blocking=True)
            ready[i].record(copy)

    for e in done:
        e.record()
    upload(0)
    t0 = None
    for s in range(warmup + steps):
        if s == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        i = s & 1
        upload(i ^ 1)                           # next step's inputs travel while this step computes
        torch.cuda.current_stream().wait_event(ready[i])
        bench.one_step(pipe, bufs[i][0], bufs[i][1], clips)
        done[i].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mb = (h_frames.numel() + h_wav.numel() * 4) / 1e6
    print(f"host-resident inputs: {clips * steps / dt:.1f} clips/s, {dt / steps * 1e3:.1f} ms/step, {mb:.0f} MB over PCIe per step")
