#!/usr/bin/env python3
"""Is the 128-clip step launch-bound?  Captures one whole step (both HIP streams) into a HIP graph and replays it against the
eager step.  Round 4, MI355X: 60.69 ms against 60.68 ms -- the capture works (identical outputs), and there is nothing to gain:
the card is busy and at its power cap, the host runs far ahead of it.  (profiles/experiments/README.md)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from avcer_amd import synth
from avcer_amd.pipeline import AVPipeline
pipe = AVPipeline(device=0, seed=42)
pipe.overlap_branches = True
dev = pipe.engine.device
frames = torch.from_numpy(synth.face_frames(1234, 128 * 16)).reshape(128, 16, 224, 224, 3).to(dev)
wav = torch.from_numpy(synth.waveforms(5678, 128, 32000)).to(dev)
def step():
    return pipe.run_clips(frames, wav, 25)
for _ in range(3): out = step()
torch.cuda.synchronize()
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("eager  %.2f ms/step" % timeit(step))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        gout = step()
    torch.cuda.synchronize()
    print("graph  %.2f ms/step" % timeit(g.replay))
    print("eager  %.2f ms/step" % timeit(step))
    print("graph  %.2f ms/step" % timeit(g.replay))
    ref = step(); torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    print("same compound argmax:", bool((ref["compound_argmax"] == gout["compound_argmax"]).all()), " max|dprob|", float((ref["compound_prob"] - gout["compound_prob"]).abs().max()))
except Exception as e:
    print("capture failed:", repr(e)[:600])
