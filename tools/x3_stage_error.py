#!/usr/bin/env python3
"""Per-stage error of the static CNN in the split-fp16 (x3) and f32 modes against the CPU oracle, over weight seeds.

For every seed: 8 frames, the stage taps (stem, layer1-4, avgpool), the fc1 features and the logits; printed as
rms(err) / rms(ref) per tap and max |dlogit| -- where the x3 logit error is made, and how it compares with the f32
MFMA path's own distance from the oracle.  Uses oracle/ as the checker (a tool, not product).

    python tools/x3_stage_error.py [seed ...]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avcer_amd import synth  # noqa: E402
from avcer_amd.engine import MODE_F16X3, MODE_FP32, Engine  # noqa: E402
from avcer_amd.sp32 import raw_to_f32  # noqa: E402
from oracle import video as ov  # noqa: E402

TAPS = ("stem", "layer1", "layer2", "layer3", "layer4")


def sp32_to_f32(raw_i16, shape):
    return raw_to_f32(raw_i16, shape)


def main():
    args = sys.argv[1:]
    if args and args[0] == "--lib":  # a one-off experiment build of the library
        from avcer_amd import _lib
        _lib.LIB = os.path.abspath(args[1])
        args = args[2:]
    seeds = [int(a) for a in args] or [42, 43, 44, 45, 46]
    eng = Engine(0)
    frames = synth.face_frames(2468, 8)
    ft = torch.from_numpy(frames)
    print("seed mode   " + " ".join(f"{t:>9}" for t in TAPS + ("avgpool", "feats")) + "   max|dlogit|  max|dprob|")
    for seed in seeds:
        sd = synth.static_state_dict(seed)
        eng.load_static(sd)
        tsd = synth.to_torch(sd)
        taps = {}
        with torch.no_grad():
            logits, feats = ov.resnet50_forward(tsd, ov.pth_processing(frames), taps)
            probs = torch.softmax(logits, 1)
        for name, mode in (("fp32", MODE_FP32), ("x3", MODE_F16X3)):
            rel = []
            for t in TAPS:
                ref = taps[t].permute(0, 2, 3, 1).contiguous()
                if t in ("layer1", "layer2", "layer3"):
                    ref = ref[:, ::2, ::2].contiguous()  # the library evaluates a stage's last block where the next stage reads it
                if mode == MODE_F16X3:
                    dst = eng.debug_tap(t, ref.numel() * 2, dtype=torch.int16)
                else:
                    dst = eng.debug_tap(t, ref.numel())
                eng.static_forward(ft, mode)
                torch.cuda.synchronize()
                got = sp32_to_f32(dst.cpu(), ref.shape) if mode == MODE_F16X3 else dst.cpu().view(ref.shape)
                rel.append(((got - ref).double().pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt()).item())
            dst = eng.debug_tap("avgpool", taps["avgpool"].numel())
            lg, pr, fe = eng.static_forward(ft, mode)
            torch.cuda.synchronize()
            ap = dst.cpu().view(taps["avgpool"].shape)
            rel.append(((ap - taps["avgpool"]).double().pow(2).mean().sqrt() / taps["avgpool"].double().pow(2).mean().sqrt()).item())
            rel.append(((fe.cpu() - feats).double().pow(2).mean().sqrt() / feats.double().pow(2).mean().sqrt()).item())
            print(f"{seed:4d} {name:5s}  " + " ".join(f"{r:9.2e}" for r in rel) +
                  f"   {(lg.cpu() - logits).abs().max().item():.3e}    {(pr.cpu() - probs).abs().max().item():.3e}"
                  f"   (logit rms {logits.pow(2).mean().sqrt().item():.2f})")


if __name__ == "__main__":
    main()
