"""Pins oracle/video.py against vectors produced by the imported reference (tests/golden/make_golden.py)."""
import numpy as np
import torch

from avcer_amd import synth
from oracle import video as ov


def _stats(t):
    t = t.detach().float()
    return np.array([t.mean().item(), t.abs().max().item(), t.std().item()])


def test_preprocess_matches_reference(golden):
    g = golden("static")
    x = ov.pth_processing(synth.face_frames(1234, 8))
    np.testing.assert_array_equal(x.reshape(-1)[:16].numpy(), g["pre_head16"])
    np.testing.assert_allclose(_stats(x), g["pre_stats"], rtol=1e-6)


def test_nearest_resize_matches_pil(golden):
    g = golden("static")
    odd = synth.u8(77, "odd", tuple(g["resize_in_shape"]))
    x = ov.pth_processing(ov.nearest_resize_u8(odd)[None])
    np.testing.assert_array_equal(x[0, :, ::16, ::16].numpy(), g["resize_out"])


def test_resnet50_matches_reference(golden, sd_static):
    g = golden("static")
    taps = {}
    with torch.no_grad():
        logits, feats = ov.resnet50_forward(sd_static, ov.pth_processing(synth.face_frames(1234, 8)), taps)
        probs = torch.softmax(logits, dim=1)
    for k in ("stem", "layer1", "layer2", "layer3", "layer4", "avgpool"):
        np.testing.assert_allclose(taps[k].reshape(-1)[:16].numpy(), g[f"{k}_head16"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(_stats(taps[k]), g[f"{k}_stats"], rtol=1e-5)
    np.testing.assert_allclose(feats.numpy(), g["feats"], atol=2e-5)
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=2e-5)
    assert np.abs(probs.numpy() - g["probs"]).max() < 1e-6
    assert (probs.argmax(1).numpy() == g["probs"].argmax(1)).all()


def test_lstm_matches_reference(golden, sd_dynamic):
    w = np.maximum(synth.centered(5, "lstm_in", (4, 10, 512), 1.0), 0).astype(np.float32)
    w[0] = w[0, 0]
    with torch.no_grad():
        lo = ov.lstm_forward(sd_dynamic, torch.from_numpy(w))
    assert np.abs(lo.numpy() - golden("lstm")["logits"]).max() < 1e-6


def test_visual_harness_matches_reference(golden, sd_static, sd_dynamic):
    g = golden("visual_harness")
    clip = synth.face_frames(4321, 16)
    for name in ("gap25", "gap30", "lead25", "full25"):
        present = g[f"{name}_present"]
        fps = float(g[f"{name}_fps"])
        st, dy = ov.visual_forward(sd_static, sd_dynamic, clip, present, fps)
        assert st.dtype == g[f"{name}_static"].dtype and dy.dtype == g[f"{name}_dynamic"].dtype
        assert np.abs(st - g[f"{name}_static"]).max() < 5e-6
        assert np.abs(dy - g[f"{name}_dynamic"]).max() < 1e-5
        # batched CNN evaluation is the same function up to reduction order
        st_b, dy_b = ov.visual_forward(sd_static, sd_dynamic, clip, present, fps, batched=True)
        assert np.abs(st_b - st).max() < 1e-5 and np.abs(dy_b - dy).max() < 1e-4


def test_lstm_step_rounding():
    assert [ov.lstm_step(f) for f in (24, 25, 29, 30, 60, 12.5)] == [5, 5, 6, 6, 12, 2]
