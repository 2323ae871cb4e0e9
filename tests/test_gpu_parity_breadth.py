"""Parity margin of the two parity-grade arithmetic modes over several synthetic weight draws and logit scales.

Real checkpoints are not available offline, so the 1e-4 probability gate is exercised on 5 weight seeds x logit scale
{1, 4, 8} (the last Linear of every model scaled: a sharper softmax makes the probabilities that many times more
sensitive to a logit error -- trained emotion heads are routinely sharper than the synthetic generator's) at 8 frames + 2
audio windows, in the f32 mode and in the split-fp16 (x3) headline mode.  Every case is printed.

ONE gate for both modes at EVERY scale: 1e-4 (north_star).  History: the x3 mode of rounds 1-3 split its operands into
bf16 pairs (4.5e-6 relative error per contraction); it sat at 1.5-2.1e-5 at scale 1, 4.2-6.6e-5 at scale 4 and AT the gate
at scale 8 (6.6e-5 ... 1.02e-4 over three arithmetically equivalent builds; 1.9e-4 in a 16-seed sweep), and round 3 bounded
scale 8 by 2e-4.  Round 4 splits into fp16 pairs on v_mfma_f32_16x16x32_f16 (csrc/split_dev.h: 11 + 11 significand bits,
7e-8 per contraction, weights pre-scaled by a power of two per matrix): the x3 taps now carry the f32 mode's own error
(layer4 1.5e-6 relative in both, tools/x3_stage_error.py), so the two modes sit at the same distance from the oracle at
every scale and the loosened branch is gone.  tools/x3_margin_sweep.py --all repeats it over 48 further draws of all three
models (profiles/r04_x3_margin_sweep_48seeds_all_models.txt: x3 worst 3.9e-6 / 6.9e-6 / 1.1e-5, f32 mode 3.9e-6 / 8.5e-6 / 1.2e-5)."""
import numpy as np
import pytest
import torch

from avcer_amd import synth
from avcer_amd.engine import MODE_F16X3, MODE_FP32, Engine
from oracle import audio as oa
from oracle import fusion as of
from oracle import video as ov

pytestmark = pytest.mark.gpu

SEEDS = (42, 43, 44, 45, 46)
SCALES = (1.0, 4.0, 8.0)


def _scaled(sd, keys, s):
    sd = dict(sd)
    for k in keys:
        sd[k] = sd[k] * s
    return sd


def test_probability_gate_over_seeds_and_logit_scales():
    eng = Engine(0)
    frames = synth.face_frames(2468, 8)
    wav = synth.waveforms(1357, 2, 32000)
    worst = {(m, sc): 0.0 for m in ("fp32", "x3") for sc in SCALES}
    rows = []
    for seed in SEEDS:
        base = (synth.static_state_dict(seed), synth.dynamic_state_dict(seed), synth.audio_state_dict(seed))
        for scale in SCALES:
            sds = (_scaled(base[0], ("fc2.weight", "fc2.bias"), scale), _scaled(base[1], ("fc.weight", "fc.bias"), scale),
                   _scaled(base[2], ("feature_downsample.weight", "feature_downsample.bias"), scale))
            eng.load_static(sds[0]); eng.load_dynamic(sds[1]); eng.load_audio(sds[2])
            tsd = [synth.to_torch(s) for s in sds]
            with torch.no_grad():
                ref_logits, _ = ov.resnet50_forward(tsd[0], ov.pth_processing(frames))
                ref_p = torch.softmax(ref_logits, 1).numpy()
                ref_a = oa.expr_model_v3_forward(tsd[2], torch.from_numpy(oa.normalize(wav))).numpy()
            for name, mode in (("fp32", MODE_FP32), ("x3", MODE_F16X3)):
                _, probs, feats = eng.static_forward(torch.from_numpy(frames), mode)
                d_s = float(np.abs(probs.cpu().numpy() - ref_p).max())
                # LSTM on the GPU's own features of the 8 frames (window = first frame x10 sliding), against the oracle LSTM
                win = torch.relu(feats.cpu())[[0] * 9 + [0], :][None].repeat(2, 1, 1)
                win[1] = torch.relu(feats.cpu())[[0, 0, 0, 1, 2, 3, 4, 5, 6, 7]]
                with torch.no_grad():
                    ref_d = torch.softmax(ov.lstm_forward(tsd[1], win), 1).numpy()
                got_d = torch.softmax(eng.dynamic_forward(win, mode).cpu(), 1).numpy()
                d_d = float(np.abs(got_d - ref_d).max())
                got_a = eng.audio_forward(torch.from_numpy(wav), True, mode).cpu().numpy()
                d_a = float(np.abs(of.softmax(got_a[:, :7]) - of.softmax(ref_a[:, :7])).max())
                assert (probs.cpu().numpy().argmax(1) == ref_p.argmax(1)).all() or d_s < 1e-6, (seed, scale, name)
                worst[(name, scale)] = max(worst[(name, scale)], d_s, d_d, d_a)
                rows.append((seed, scale, name, d_s, d_d, d_a))
    for r in rows:
        print("seed %d scale %.0f %-4s  static %.2e  dynamic %.2e  audio %.2e" % r)
    for sc in SCALES:
        print("worst |dprob| over %d seeds at logit scale %.0f: fp32 %.3e, x3 %.3e" % (len(SEEDS), sc, worst[("fp32", sc)], worst[("x3", sc)]))
    for sc in SCALES:
        assert worst[("fp32", sc)] < 1e-4, (sc, worst)
        assert worst[("x3", sc)] < 1e-4, (sc, worst)
    eng.close()


def test_config2_static_batch_256_in_every_arithmetic_mode(engine_static, sd_static):
    """BASELINE configs[1] at its stated size: the static CNN on 256 frames in ONE call, in bf16 (the dtype the config
    names), x3 and f32.  32 rows spread over the batch are checked against the CPU oracle: the two parity-grade modes at
    the 1e-4 gate, plain bf16 at its own measured bound (it misses the gate by two orders of magnitude and is never the
    headline); every mode must keep the oracle's argmax wherever the oracle's top-2 margin exceeds the mode's error."""
    from avcer_amd.engine import MODE_BF16

    frames = synth.face_frames(1357, 256)
    idx = np.arange(0, 256, 8)
    with torch.no_grad():
        ref_logits, _ = ov.resnet50_forward(sd_static, ov.pth_processing(frames[idx]))
        ref = torch.softmax(ref_logits, 1).numpy()
    for name, mode, tol in (("fp32", MODE_FP32, 1e-4), ("x3", MODE_F16X3, 1e-4), ("bf16", MODE_BF16, 5e-2)):
        _, probs, _ = engine_static.static_forward(torch.from_numpy(frames), mode)
        p = probs.cpu().numpy()[idx]
        d = float(np.abs(p - ref).max())
        print("config 2 (batch 256) %-4s max|dprob| over 32 sampled rows %.3e" % (name, d))
        assert np.isfinite(p).all() and d < tol, (name, d)
        top2 = np.sort(ref, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 2 * d
        assert (p.argmax(1)[clear] == ref.argmax(1)[clear]).all(), name
