"""GPU parity of the fusion kernels and of the whole AV path against the oracle / golden vectors."""
import numpy as np
import pytest
import torch

from avcer_amd import fusion, synth
from avcer_amd.engine import MODE_FP32
from oracle import audio as oa
from oracle import fusion as of
from oracle import video as ov

pytestmark = pytest.mark.gpu


def _spans_from_frames(rows, frames):
    """Recover windows (one per distinct logit row run) from the golden's replicated rows."""
    wins, lo, hi = [], [], []
    i = 0
    while i < len(frames):
        j = i
        while j + 1 < len(frames) and frames[j + 1] == frames[j] + 1 and np.array_equal(rows[j + 1], rows[i]):
            j += 1
        wins.append(rows[i]); lo.append(frames[i]); hi.append(frames[j] + 1)
        i = j + 1
    return np.stack(wins), np.array(lo), np.array(hi)


def test_fuse_matches_reference(engine, golden):
    g = golden("fusion")
    for c in range(int(g["n_cases"])):
        stat, dyn = g[f"c{c}_stat"], g[f"c{c}_dyn"]
        wins, lo, hi = _spans_from_frames(g[f"c{c}_aud_rows"], g[f"c{c}_aud_frames"])
        for wname, w1 in (("w", fusion.WEIGHTS_AV_1), ("none", None)):
            for cwt in (False, True):
                for cm in (False, True):
                    key = f"c{c}_{wname}_{int(cwt)}{int(cm)}"
                    prob, am = fusion.fuse(engine, stat, dyn, wins, lo, hi, w1, (1, 1, 1), cwt, cm)
                    np.testing.assert_allclose(prob.cpu().numpy(), g[key + "_prob"], rtol=0, atol=2e-6)
                    ref_am = g[key + "_argmax"]
                    got_am = am.cpu().numpy()
                    if not np.array_equal(got_am, ref_am):  # only exact ties / 1-ulp neighbours may differ
                        p = g[key + "_prob"]
                        bad = got_am != ref_am
                        top = np.take_along_axis(p, ref_am[..., None], 2)[..., 0]
                        alt = np.take_along_axis(p, got_am[..., None].astype(np.int64), 2)[..., 0]
                        assert np.abs(top - alt)[bad].max() < 1e-6, key


def test_frame_mean(engine):
    wins = synth.centered(1, "fm", (5, 8), 1.0)
    lo, hi = np.array([0, 3, 6, 9, 40]), np.array([8, 11, 14, 17, 50])
    mean, cnt = engine.audio_frame_mean(wins, lo, hi, 20)
    rows, frames = oa.replicate_per_frame(wins, [(0, 0, a, b) for a, b in zip(lo, hi)])
    for f in range(20):
        sel = rows[frames == f]
        assert int(cnt[f]) == len(sel)
        if len(sel):
            np.testing.assert_allclose(mean[f].cpu().numpy(), sel.mean(0), atol=1e-6)


def test_full_av_clips_match_oracle(engine, sd_static, sd_dynamic, sd_audio):
    """BASELINE config 4 at a size the oracle finishes in seconds: 3 clips of 16 frames + 2 s audio, all stages."""
    from avcer_amd.pipeline import AVPipeline

    pipe = AVPipeline.__new__(AVPipeline)
    pipe.engine, pipe.mode = engine, MODE_FP32
    engine.load_static(sd_static); engine.load_dynamic(sd_dynamic); engine.load_audio(sd_audio)
    n, t = 3, 16
    frames = synth.face_frames(2024, n * t).reshape(n, t, 224, 224, 3)
    wav = synth.waveforms(2025, n, 32000)
    present = np.ones((n, t), bool)
    present[1, 3:6] = False
    out = pipe.run_clips(torch.from_numpy(frames), torch.from_numpy(wav), 25, present)
    for c in range(n):
        st, dy = ov.visual_forward(sd_static, sd_dynamic, frames[c], present[c], 25, batched=True)
        with torch.no_grad():
            lg = oa.expr_model_v3_forward(sd_audio, torch.from_numpy(oa.normalize(wav[c:c + 1]))).numpy().reshape(1, 8)
        rows, fr = oa.replicate_per_frame(lg, [(0, 32000, 0, t)])
        prob, am = of.fuse(st.astype(np.float32), dy.astype(np.float32), rows, fr)
        assert np.abs(out["static_probs"][c].cpu().numpy() - st).max() < 1e-4
        print("clip", c, "max|d dynamic logit|", np.abs(out["dynamic_logits"][c].cpu().numpy() - dy).max())
        assert np.abs(out["dynamic_logits"][c].cpu().numpy() - dy).max() < 1e-5  # measured 1.1e-6
        assert np.abs(out["compound_prob"][:, c].cpu().numpy() - prob).max() < 1e-4
        got_am = out["compound_argmax"][:, c].cpu().numpy()
        if not np.array_equal(got_am, am):
            bad = got_am != am
            top = np.take_along_axis(prob, am[..., None], 2)[..., 0]
            alt = np.take_along_axis(prob, got_am[..., None].astype(np.int64), 2)[..., 0]
            assert np.abs(top - alt)[bad].max() < 1e-4


def test_full_size_batch_properties_x3(engine, sd_static, sd_dynamic, sd_audio):
    """BASELINE config 4/5 per-GPU size (128 clips = 2048 frames + 128 windows) in the headline arithmetic mode:
    determinism, independence of batch composition (a clip's record does not depend on its neighbours or position),
    probabilities normalised, and agreement of a sample of clips with the oracle within the 1e-4 gate."""
    from avcer_amd.engine import MODE_F16X3
    from avcer_amd.pipeline import AVPipeline

    pipe = AVPipeline.__new__(AVPipeline)
    pipe.engine, pipe.mode = engine, MODE_F16X3
    engine.load_static(sd_static); engine.load_dynamic(sd_dynamic); engine.load_audio(sd_audio)
    n, t = 128, 16
    frames = torch.from_numpy(synth.face_frames(31337, n * t).reshape(n, t, 224, 224, 3)).to(engine.device)
    wav = torch.from_numpy(synth.waveforms(31338, n, 32000)).to(engine.device)
    a = pipe.run_clips(frames, wav, 25)
    b = pipe.run_clips(frames, wav, 25)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1)).to(engine.device)
    c = pipe.run_clips(frames[perm], wav[perm], 25)
    assert torch.equal(c["static_probs"], a["static_probs"][perm])
    assert torch.equal(c["dynamic_logits"], a["dynamic_logits"][perm])
    assert torch.equal(c["audio_logits"], a["audio_logits"][perm])
    assert torch.equal(c["compound_argmax"], a["compound_argmax"][:, perm])
    sub = pipe.run_clips(frames[40:43], wav[40:43], 25)
    assert torch.equal(sub["static_probs"], a["static_probs"][40:43])
    assert torch.equal(sub["audio_logits"], a["audio_logits"][40:43])
    p = a["static_probs"]
    assert torch.isfinite(p).all() and (p.sum(-1) - 1).abs().max() < 1e-5
    for c_ in (0, 77, 127):
        st, dy = ov.visual_forward(sd_static, sd_dynamic, frames[c_].cpu().numpy(), np.ones(t, bool), 25, batched=True)
        assert np.abs(a["static_probs"][c_].cpu().numpy() - st).max() < 1e-4
        with torch.no_grad():
            lg = oa.expr_model_v3_forward(sd_audio, torch.from_numpy(oa.normalize(wav[c_:c_ + 1].cpu().numpy()))).numpy().reshape(1, 8)
        d = np.abs(of.softmax(a["audio_logits"][c_:c_ + 1, :7].cpu().numpy()) - of.softmax(lg[:, :7])).max()
        assert d < 1e-4
