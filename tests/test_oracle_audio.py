"""Pins oracle/audio.py against vectors produced by the imported reference + installed transformers."""
import numpy as np
import pytest
import torch

from avcer_amd import synth
from oracle import audio as oa


def _stats(t):
    t = t.detach().float()
    return np.array([t.mean().item(), t.abs().max().item(), t.std().item()])


def test_padding_and_normaliser(golden):
    g = golden("audio_pad")
    win = 4000
    for n in (1, 999, 4000, 4001):
        wav = torch.from_numpy(synth.waveforms(9, 1, n)[0])
        for mode in ("mean", "constant"):
            p = oa.pad_wav_zeros(wav, win, mode=mode)
            np.testing.assert_array_equal(p.numpy(), g[f"pad_{mode}_{n}"])
            got = oa.normalize(p.unsqueeze(0).numpy())
            assert got.shape == g[f"norm_{mode}_{n}"].shape == (1, max(win, n))
            np.testing.assert_allclose(got, g[f"norm_{mode}_{n}"], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(oa.pad_wav(wav, win).numpy(), g[f"pad_repeat_{n}"])
    empty = oa.pad_wav_zeros(torch.zeros(0), 8, mode="mean")
    assert np.isnan(empty.numpy()).all() and np.isnan(g["pad_mean_0"]).all()


def test_chunker_frame_mapping(golden):
    g = golden("chunker")

    def probe(x):
        return np.stack([x[:, 0], x[:, -1], x[:, x.shape[1] // 2], x[:, 123], x.mean(1), np.abs(x).max(1),
                         x[:, 1], x[:, -2]], axis=1)

    keys = sorted(k[:-7] for k in g.files if k.endswith("_frames"))
    assert len(keys) >= 50
    for key in keys:
        fps, n, w, s, padding = key.split("_")
        fps, n, w, s = int(fps[3:]), int(n[1:]), int(w[1:]), float(s[1:])
        wav = torch.from_numpy(synth.waveforms(31, 1, n)[0])
        chunks, spans = oa.make_chunks(wav, 16000, fps, w, s, padding)
        rows, frames = oa.replicate_per_frame(probe(chunks), spans)
        np.testing.assert_array_equal(frames, g[key + "_frames"])
        np.testing.assert_allclose(rows, g[key + "_rows"], rtol=0, atol=2e-6, equal_nan=True)


def test_empty_tail_chunk_is_nan():
    wav = torch.from_numpy(synth.waveforms(1, 1, 16000)[0])
    chunks, spans = oa.make_chunks(wav, 16000, 25, 4, 0.5, "mean")
    assert spans[-1][0] == spans[-1][1] == 16000
    assert np.isnan(chunks[-1]).all() and not np.isnan(chunks[:-1]).any()


@pytest.mark.parametrize("tag,seed,b,t", [("t32000", 5678, 2, 32000), ("t64000", 5679, 1, 64000)])
def test_expr_model_v3_matches_reference(golden, sd_audio, tag, seed, b, t):
    g = golden("audio_model")
    x = oa.normalize(synth.waveforms(seed, b, t))
    np.testing.assert_allclose(x.reshape(-1)[:16], g[f"{tag}_input_head16"], atol=1e-6)
    taps = {}
    with torch.no_grad():
        lg = oa.expr_model_v3_forward(sd_audio, torch.from_numpy(x), taps)
    assert tuple(lg.shape) == tuple(g[f"{tag}_logits"].shape)  # (8,) at batch 1: `.squeeze()`
    for k in ("conv0", "extract", "proj", "layer0", "layer5", "layer11", "w2v", "tl1", "tl2"):
        assert tuple(taps[k].shape) == tuple(g[f"{tag}_{k}_shape"])
        np.testing.assert_allclose(taps[k].reshape(-1)[:16].numpy(), g[f"{tag}_{k}_head16"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(_stats(taps[k]), g[f"{tag}_{k}_stats"], rtol=1e-4, atol=1e-6)
    assert np.abs(lg.numpy() - g[f"{tag}_logits"]).max() < 2e-5
    assert np.array_equal(lg.numpy().reshape(-1, 8).argmax(1), g[f"{tag}_logits"].reshape(-1, 8).argmax(1))


def test_param_count(golden, sd_audio):
    n = sum(v.numel() for k, v in sd_audio.items()
            if k.startswith("wav2vec2.") and "running" not in k and "num_batches" not in k)
    assert n == int(golden("audio_model")["n_params"][0]) == 164284032


def test_seven_class_variant_matches_reference(golden):
    """Row f3: ExprModelV2 of architectures/audio_7_cl.py (same graph, 7-wide last Linear)."""
    sd = synth.to_torch(synth.audio_state_dict(43, num_classes=7))
    x = oa.normalize(synth.waveforms(777, 2, 32000))
    with torch.no_grad():
        lg = oa.expr_model_v3_forward(sd, torch.from_numpy(x))
    assert tuple(lg.shape) == (2, 7)
    assert np.abs(lg.numpy() - golden("audio_model7")["logits"]).max() < 2e-5
