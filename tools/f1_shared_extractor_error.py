#!/usr/bin/env python3
"""Row f1 of SURVEY.md section 8 ("overlap-aware audio encoding"), bounded with a number (round-5 review item 8).  CPU, oracle only.

The reference encodes every 4 s window on its own (get_prob_audio_8_cl.py:70-92; run.py:247-251: window 4 s, step 0.5 s), so
every sample passes the 7-layer convolutional extractor 8 times.  Each window is normalised with ITS OWN mean and variance
first (HF feature extractor, a8), and the extractor's first LayerNorm (over the 512 channels of conv0's output) does not commute
with that affine: from layer 0's LayerNorm on, every activation depends on the window's (mean, std).  What CAN be shared exactly is
conv0's linear part alone -- W x, 0.07 of the extractor's 9.8 GFLOP.  Sharing more means one normalisation for all windows.  This
script measures what that approximation costs: the oracle's per-window probabilities as the reference computes them against
windows cut from ONE extractor pass over the whole clip (clip-level mean / std; extractor stride 320 samples divides the 8000-sample
window step, so window w is rows [25 w, 25 w + 199) of the shared feature map; projection, encoder and head per window as before).

    python tools/f1_shared_extractor_error.py [seconds=30]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from avcer_amd import synth  # noqa: E402
from oracle import audio as oa  # noqa: E402
from oracle import fusion as of  # noqa: E402


def tail_from_features(sd, f):
    """wav2vec2_forward from the extractor's output on (oracle/audio.py:158-172) + the first-party layers."""
    p = "wav2vec2.feature_projection."
    f = F.layer_norm(f, (f.shape[-1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], oa.LN_EPS)
    h = F.linear(f, sd[p + "projection.weight"], sd[p + "projection.bias"])
    h = oa.encoder(sd, h)
    h = oa.transformer_layer(sd, "tl1", h, 32)
    h = oa.transformer_layer(sd, "tl2", h, 16)
    return oa.head(sd, h)


def main():
    seconds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    torch.set_num_threads(os.cpu_count() or 1)
    sd = synth.to_torch(synth.audio_state_dict(42))
    sr, win, step = 16000, 4 * 16000, 8000
    base = synth.waveforms(78, 1, seconds * sr)[0]
    t = np.arange(len(base)) / sr
    cases = {
        "stationary noise (the bench's waveform)": base,
        "loudness ramp x10 over the clip": (base * (0.1 + 0.9 * t / t[-1])).astype(np.float32),
        "speech-like bursts (0.3 s on / 0.5 s off, +0.02 DC drift)": (base * (np.sin(2 * np.pi * t / 0.8) > 0.2) + 0.02 * np.sin(2 * np.pi * t / 7.0)).astype(np.float32),
    }
    n_full = (len(base) - win) // step + 1
    for name, wav in cases.items():
        with torch.no_grad():
            ref = np.stack([oa.expr_model_v3_forward(sd, torch.from_numpy(oa.normalize(wav[None, w * step:w * step + win]))).numpy().reshape(-1)
                            for w in range(n_full)])
            feats = oa.feature_extractor(sd, torch.from_numpy(oa.normalize(wav[None])))      # ONE pass, clip-level normalisation
            assert feats.shape[1] >= 25 * (n_full - 1) + 199
            got = np.stack([tail_from_features(sd, feats[:, 25 * w:25 * w + 199]).numpy().reshape(-1) for w in range(n_full)])
            # the same windows with the extractor run per window but normalised with the CLIP's statistics: separates the effect of
            # the normalisation from any effect of where the window is cut (must agree with `got` to rounding)
            mu, sd_ = wav.mean(), np.sqrt(wav.var() + 1e-7)
            chk = np.stack([oa.expr_model_v3_forward(sd, torch.from_numpy(((wav[None, w * step:w * step + win] - mu) / sd_).astype(np.float32))).numpy().reshape(-1)
                            for w in range(n_full)])
        p_ref, p_got, p_chk = of.softmax(ref[:, :7]), of.softmax(got[:, :7]), of.softmax(chk[:, :7])
        d = np.abs(p_got - p_ref).max(axis=1)
        stats = np.array([[wav[w * step:w * step + win].mean(), wav[w * step:w * step + win].std()] for w in range(n_full)])
        print(f"\n{name}: {n_full} full windows of 4 s every 0.5 s")
        print(f"  per-window std / clip std: {stats[:, 1].min() / wav.std():.4f} .. {stats[:, 1].max() / wav.std():.4f}; per-window mean / clip std: "
              f"{np.abs(stats[:, 0]).max() / wav.std():.2e}")
        print(f"  max |dprob| per window, shared extractor vs the reference's per-window encoding: median {np.median(d):.2e}, max {d.max():.2e} "
              f"(window {int(d.argmax())}); argmax changed in {int((p_got.argmax(1) != p_ref.argmax(1)).sum())} of {n_full} windows")
        print(f"  (cut from the shared map vs extractor per window under the same clip-level normalisation: max |dprob| {np.abs(p_got - p_chk).max():.2e})")
    print("\nextractor FLOPs shareable exactly (conv0's linear part): 0.0655 of 9.809 GFLOP per 2 s window = 0.7 %; sharing layers 0-6 "
          "(-20 % of the audio model's FLOPs at 8 x overlap) needs one normalisation for all windows: the error above")


if __name__ == "__main__":
    main()
