#!/usr/bin/env python3
"""Would folding LayerNorm(1024) into the GEMM behind it stay f32-grade?  (CPU only; numpy emulation, no GPU.)

The 28 LN(1024) launches in front of qkv / ffn1 of the wav2vec2 layers cost 0.85 ms of a 60 ms step.  The fold the round-4
review sketches:  y = LN(h) W^T + b  with  LN(h) = (h - mu) / sigma * gamma + beta  becomes

    y = rstd * [ h (gamma * W)^T  -  mu * c ]  +  d,      c = (gamma * W) 1,   d = beta W^T + b

-- the GEMM runs on the RAW residual stream h, gamma lives in the weights, the row statistics (sum x, sum x^2, accumulated in
f32 by the epilogue of the GEMM that PRODUCED h) enter in the consumer's epilogue.  Two cancellations decide whether that is
usable: `h W'^T - mu c` subtracts two numbers of size |mu| |W'| sqrt(K) to leave one of size sigma |W'| sqrt(K), and
`E[x^2] - mu^2` does the same to the variance.  Both amplify the f32 rounding of their operands by about (rms(h) / sigma)^2
resp. rms(h) / sigma.  This script measures it on the oracle's residual streams (synthetic weights: the only ones in the
image): relative rms error against a float64 evaluation of (a) the shipped order, LN then GEMM, in f32, (b) the folded order
in f32 with one-pass statistics, (c) the folded order with exact (two-pass) statistics.

    python tools/ln_fold_error.py [seed ...]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avcer_amd import synth  # noqa: E402
from oracle import audio as oa  # noqa: E402

EPS = 1e-5


def rel(a, ref):
    return float(np.sqrt(np.mean((a.astype(np.float64) - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))


def main():
    seeds = [int(a) for a in sys.argv[1:]] or [42, 43]
    wav = synth.waveforms(5678, 2, 32000)
    print("seed stream   layer  rms(h)/sigma (median / max over rows)   shipped f32     folded, one-pass stats   folded, exact stats   "
          "folded with an offset of 20 sigma added to every row")
    for seed in seeds:
        sd = synth.to_torch(synth.audio_state_dict(seed))
        taps = {}
        with torch.no_grad():
            oa.expr_model_v3_forward(sd, torch.from_numpy(oa.normalize(wav)), taps)
        for tap, layer in (("posconv", 0), ("layer0", 1), ("layer5", 6), ("layer11", 11)):
            h = taps[tap].reshape(-1, 1024).numpy().astype(np.float32)
            p = f"wav2vec2.encoder.layers.{layer}"
            g = sd[p + ".layer_norm.weight"].numpy().astype(np.float64)
            be = sd[p + ".layer_norm.bias"].numpy().astype(np.float64)
            w = torch.cat([sd[p + f".attention.{n}_proj.weight"] for n in "qkv"]).numpy().astype(np.float64)   # [3072, 1024]

            def run(hh):
                h64 = hh.astype(np.float64)
                mu, var = h64.mean(1, keepdims=True), h64.var(1, keepdims=True)
                ref = ((h64 - mu) / np.sqrt(var + EPS) * g + be) @ w.T
                # (a) the shipped order in f32
                mu32 = hh.mean(1, keepdims=True, dtype=np.float32)
                d32 = hh - mu32
                r32 = (1.0 / np.sqrt((d32 * d32).mean(1, keepdims=True, dtype=np.float32) + np.float32(EPS))).astype(np.float32)
                ln32 = (d32 * r32 * g.astype(np.float32) + be.astype(np.float32)).astype(np.float32)
                a = ln32 @ w.T.astype(np.float32)
                # (b) folded: f32 GEMM on the raw stream, one-pass f32 statistics
                wp = (w * g).astype(np.float32)
                c = wp.astype(np.float64).sum(1).astype(np.float32)
                d = (be @ w.T).astype(np.float32)
                z = hh @ wp.T
                s1 = hh.sum(1, keepdims=True, dtype=np.float32) / np.float32(1024)
                s2 = (hh * hh).sum(1, keepdims=True, dtype=np.float32) / np.float32(1024)
                rs = (1.0 / np.sqrt(np.maximum(s2 - s1 * s1, 0) + np.float32(EPS))).astype(np.float32)
                b = (z - s1 * c) * rs + d
                # (c) folded with exact statistics
                cex = (z - mu.astype(np.float32) * c) * (1.0 / np.sqrt(var + EPS)).astype(np.float32) + d
                ratio = np.sqrt((h64 ** 2).mean(1)) / np.sqrt(var[:, 0] + EPS)
                return ratio, rel(a, ref), rel(b, ref), rel(cex, ref)

            ratio, ea, eb, ec = run(h)
            sig = float(np.sqrt(h.astype(np.float64).var(1).mean()))
            _, _, eb20, ec20 = run((h + np.float32(20.0 * sig)).astype(np.float32))
            print(f"{seed:4d} {tap:8s} {layer:5d}  {np.median(ratio):10.2f} / {ratio.max():8.2f}              {ea:10.2e}        {eb:10.2e}              "
                  f"{ec:10.2e}            one-pass {eb20:9.2e}, exact {ec20:9.2e}")


if __name__ == "__main__":
    main()
