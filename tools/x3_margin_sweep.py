#!/usr/bin/env python3
"""The probability-gate sweep of tests/test_gpu_parity_breadth.py over MORE weight draws than the test takes (default 16
seeds x logit scales 1 / 4 / 8, static CNN, 8 frames): how far the split-fp16 mode sits from the 1e-4 gate is a statistic,
and five seeds are a small sample of it.  Prints every case and the worst / median per scale.

    python tools/x3_margin_sweep.py [first_seed] [n_seeds] [--all]      # --all: the LSTM and the audio model too
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from avcer_amd import synth  # noqa: E402
from avcer_amd.engine import MODE_F16X3, MODE_FP32, Engine  # noqa: E402
from oracle import audio as oa  # noqa: E402
from oracle import fusion as of  # noqa: E402
from oracle import video as ov  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    every = "--all" in sys.argv
    first = int(args[0]) if len(args) > 0 else 100
    n = int(args[1]) if len(args) > 1 else 16
    eng = Engine(0)
    frames = synth.face_frames(2468, 8)
    wav = synth.waveforms(1357, 2, 32000)
    res = {(m, s): [] for m in ("fp32", "x3") for s in (1.0, 4.0, 8.0)}
    for seed in range(first, first + n):
        base = synth.static_state_dict(seed)
        for scale in (1.0, 4.0, 8.0):
            sd = dict(base)
            for k in ("fc2.weight", "fc2.bias"):
                sd[k] = sd[k] * scale
            eng.load_static(sd)
            with torch.no_grad():
                ref_logits, _ = ov.resnet50_forward(synth.to_torch(sd), ov.pth_processing(frames))
                ref = torch.softmax(ref_logits, 1).numpy()
            if every:
                sdd, sda = dict(synth.dynamic_state_dict(seed)), dict(synth.audio_state_dict(seed))
                for k in ("fc.weight", "fc.bias"):
                    sdd[k] = sdd[k] * scale
                for k in ("feature_downsample.weight", "feature_downsample.bias"):
                    sda[k] = sda[k] * scale
                eng.load_dynamic(sdd)
                eng.load_audio(sda)
                with torch.no_grad():
                    ref_a = oa.expr_model_v3_forward(synth.to_torch(sda), torch.from_numpy(oa.normalize(wav))).numpy()
            for name, mode in (("fp32", MODE_FP32), ("x3", MODE_F16X3)):
                _, probs, feats = eng.static_forward(torch.from_numpy(frames), mode)
                d = float(np.abs(probs.cpu().numpy() - ref).max())
                if every:
                    win = torch.relu(feats.cpu())[[0, 0, 0, 1, 2, 3, 4, 5, 6, 7]][None]
                    with torch.no_grad():
                        ref_d = torch.softmax(ov.lstm_forward(synth.to_torch(sdd), win), 1).numpy()
                    d = max(d, float(np.abs(torch.softmax(eng.dynamic_forward(win, mode).cpu(), 1).numpy() - ref_d).max()))
                    got_a = eng.audio_forward(torch.from_numpy(wav), True, mode).cpu().numpy()
                    d = max(d, float(np.abs(of.softmax(got_a[:, :7]) - of.softmax(ref_a[:, :7])).max()))
                res[(name, scale)].append(d)
                print(f"seed {seed} scale {scale:.0f} {name:4s} {'static + LSTM + audio' if every else 'static'} max|dprob| {d:.3e}", flush=True)
    for (name, scale), v in sorted(res.items()):
        print(f"{name:4s} scale {scale:.0f}: worst {max(v):.3e}  median {statistics.median(v):.3e}  over {len(v)} seeds, "
              f"{sum(x >= 1e-4 for x in v)} at or above 1e-4")


if __name__ == "__main__":
    main()
