#!/usr/bin/env python3
"""Per-stage error of the audio model in the split-fp16 (x3) and f32 modes against the CPU oracle: rms(err) / rms(ref) of
the stage taps and max |dlogit| -- where the x3 mode's distance from the oracle is made.  The residual-stream taps are f32
in every mode; `extract` (the conv feature extractor's output) is an MFMA operand tensor: f32 in the f32 mode, sp32 (fp16
hi / lo per 32 channels) in the x3 mode, decoded here with avcer_amd.sp32.raw_to_f32 (round 4 read it as f32: garbage).

    python tools/x3_audio_stage_error.py [--lib one-off-build.so] [seed ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
if args and args[0] == "--lib":
    from avcer_amd import _lib
    _lib.LIB = os.path.abspath(args[1])
    args = args[2:]
from avcer_amd import sp32, synth  # noqa: E402
from avcer_amd.engine import MODE_F16X3, MODE_FP32, Engine  # noqa: E402
from oracle import audio as oa  # noqa: E402

TAPS = ("extract", "proj", "posconv", "layer0", "layer5", "layer11", "w2v", "tl1", "tl2")


def main():
    seeds = [int(a) for a in args] or [42, 43, 44]
    eng = Engine(0)
    wav = synth.waveforms(5678, 2, 32000)
    print("seed mode  " + " ".join(f"{t:>8}" for t in TAPS) + "   max|dlogit|")
    for seed in seeds:
        sd = synth.audio_state_dict(seed)
        eng.load_audio(sd)
        taps = {}
        with torch.no_grad():
            x = oa.normalize(wav)
            ref_logits = oa.expr_model_v3_forward(synth.to_torch(sd), torch.from_numpy(x), taps)
        for name, mode in (("fp32", MODE_FP32), ("x3", MODE_F16X3)):
            rel = []
            for t in TAPS:
                ref = taps[t]
                split = mode == MODE_F16X3 and t == "extract"  # an operand tensor: sp32 in the x3 mode
                dst = eng.debug_tap(t, ref.numel() * 2, dtype=torch.int16) if split else eng.debug_tap(t, ref.numel())
                out = eng.audio_forward(torch.from_numpy(wav), True, mode)
                torch.cuda.synchronize()
                got = sp32.raw_to_f32(dst.cpu(), ref.shape) if split else dst.cpu().view(ref.shape)
                rel.append(((got - ref).double().pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt()).item())
            print(f"{seed:4d} {name:5s} " + " ".join(f"{r:8.1e}" for r in rel) + f"   {(out.cpu() - ref_logits).abs().max().item():.3e}")


if __name__ == "__main__":
    main()
