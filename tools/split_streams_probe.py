#!/usr/bin/env python3
"""Probe: does running the visual branch as TWO half-batches on two HIP streams (two contexts), beside the audio branch
on a third, beat one visual stream + one audio stream?  (The tails of every launch would fill with the other half's blocks.)

    python tools/split_streams_probe.py [--clips 128] [--steps 10]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avcer_amd import synth  # noqa: E402
from avcer_amd.engine import MODE_F16X3  # noqa: E402
from avcer_amd.pipeline import AVPipeline  # noqa: E402
from avcer_amd.video_pipeline import visual_forward  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=128)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    p1 = AVPipeline(device=0, seed=42, mode=MODE_F16X3)
    p2 = AVPipeline(device=0, seed=42, mode=MODE_F16X3)
    frames = torch.from_numpy(synth.face_frames(1234, a.clips * 16)).reshape(a.clips, 16, 224, 224, 3).to(dev)
    wav = torch.from_numpy(synth.waveforms(5678, a.clips, 32000)).to(dev)
    present = np.ones((a.clips, 16), bool)
    h = a.clips // 2
    s_aud, s_v2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def one_stream():
        p1.overlap_branches = False
        return p1.clip_records(frames, wav, 25)

    def two_streams():
        p1.overlap_branches = True
        return p1.clip_records(frames, wav, 25)

    def three_streams():
        main_s = torch.cuda.current_stream(dev)
        s_aud.wait_stream(main_s)
        s_v2.wait_stream(main_s)
        with torch.cuda.stream(s_aud):
            aud = p1.engine.audio_forward(wav, normalize=True, mode=MODE_F16X3)
        with torch.cuda.stream(s_v2):
            st2, dy2 = visual_forward(p2.engine, frames[h:], present[h:], 25, MODE_F16X3)
        st1, dy1 = visual_forward(p1.engine, frames[:h], present[:h], 25, MODE_F16X3)
        main_s.wait_stream(s_aud)
        main_s.wait_stream(s_v2)
        return torch.cat([st1, st2]), torch.cat([dy1, dy2]), aud

    def audio_halves():
        main_s = torch.cuda.current_stream(dev)
        s_aud.wait_stream(main_s)
        s_v2.wait_stream(main_s)
        with torch.cuda.stream(s_aud):
            a1 = p1.engine.audio_forward(wav[:h], normalize=True, mode=MODE_F16X3)
        with torch.cuda.stream(s_v2):
            a2 = p2.engine.audio_forward(wav[h:], normalize=True, mode=MODE_F16X3)
        st, dy = visual_forward(p1.engine, frames, present, 25, MODE_F16X3)
        main_s.wait_stream(s_aud)
        main_s.wait_stream(s_v2)
        return st, dy, torch.cat([a1, a2])

    ref = [t.cpu() for t in one_stream()]
    for name, fn in (("1 stream", one_stream), ("2 streams (audio | visual)", two_streams),
                     ("3 streams (audio | visual half | visual half)", three_streams),
                     ("3 streams (audio half | audio half | visual)", audio_halves)) * 2:
        out = [t.cpu() for t in fn()]
        same = all(torch.equal(x, y) for x, y in zip(ref, out))
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f"{name:48s} {dt * 1e3:7.2f} ms/step  {a.clips / dt:7.1f} clips/s  identical to 1 stream: {same}")


if __name__ == "__main__":
    main()
