"""RetinaFace-R50 detector on the GPU (row f4) against vectors produced by the reference's RetinaFace class
(tests/golden/face_net.npz) and against the CPU oracle on other frame sizes."""
import numpy as np
import pytest
import torch

from avcer_amd import face_tiles as ft
from avcer_amd import synth
from avcer_amd.engine import MODE_BF16, MODE_F16X3, MODE_FP32
from oracle import face as of
from oracle import retina as orf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd_retina():
    return synth.to_torch(synth.retina_state_dict(42))


@pytest.fixture(scope="module")
def engine_face(engine, sd_retina):
    engine.load_face(sd_retina)
    return engine


@pytest.mark.parametrize("name", ["a", "b"])
@pytest.mark.parametrize("mode,tol", [(MODE_FP32, 1e-4), (MODE_F16X3, 1e-4)])
def test_network_matches_reference_class(engine_face, golden, name, mode, tol):
    g = golden("face_net")
    h, w = (int(v) for v in g[f"{name}_size"])
    frame = synth.video_frames(900, 1, h, w)
    loc, conf, lm = (t[0].cpu().numpy() for t in engine_face.face_forward(frame, mode))
    assert loc.shape == g[f"{name}_loc"].shape and conf.shape == g[f"{name}_conf"].shape
    print(name, mode, "max|dloc|", np.abs(loc - g[f"{name}_loc"]).max(), "max|dconf|", np.abs(conf - g[f"{name}_conf"]).max(),
          "max|dlandm|", np.abs(lm - g[f"{name}_landms"]).max())
    assert np.abs(conf - g[f"{name}_conf"]).max() < tol            # probabilities: the 1e-4 gate in f32
    assert np.abs(loc - g[f"{name}_loc"]).max() < 10 * tol         # raw regression outputs, |x| up to ~10
    assert np.abs(lm - g[f"{name}_landms"]).max() < 10 * tol


def test_bf16_mode_is_close_but_not_parity(engine_face, golden):
    g = golden("face_net")
    frame = synth.video_frames(900, 1, 96, 128)
    _, conf, _ = engine_face.face_forward(frame, MODE_BF16)
    d = np.abs(conf[0].cpu().numpy() - g["a_conf"]).max()
    print("bf16 max|dconf|", d)
    assert d < 0.1


def test_batch_and_rgb_against_oracle(engine_face, sd_retina):
    frames = synth.video_frames(77, 3, 150, 214)                    # odd sizes, three frames in one call
    loc, conf, lm = engine_face.face_forward(frames, MODE_FP32)
    for i in range(3):
        rl, rc, rm = orf.retina_forward(sd_retina, orf.preprocess(frames[i]))
        assert np.abs(conf[i].cpu().numpy() - rc[0].numpy()).max() < 1e-4
        assert np.abs(loc[i].cpu().numpy() - rl[0].numpy()).max() < 1e-3
        assert np.abs(lm[i].cpu().numpy() - rm[0].numpy()).max() < 1e-3
    l2, c2, _ = engine_face.face_forward(np.ascontiguousarray(frames[:1, :, :, ::-1]), MODE_FP32, rgb=True)
    np.testing.assert_array_equal(c2[0].cpu().numpy(), conf[0].cpu().numpy())
    assert engine_face.lib.avcer_face_num_priors(150, 214) == len(of.prior_boxes((150, 214))) == loc.shape[1]


@pytest.mark.parametrize("h,w,n", [(150, 214, 3), (70, 33, 2), (75, 101, 2), (360, 640, 2)])
def test_x3_fused_body_against_oracle_and_fp32_mode(engine_face, sd_retina, h, w, n):
    """The x3 mode runs the body's stride-1 bottlenecks on the fused chain / tail kernels of the recognition CNN (api.hip
    face_forward_impl): odd extents (38 x 54 and 19 x 27 positions per frame in stages 1-2), a frame narrower than one tile row, and
    the bench's 640 x 360 -- against the oracle where it finishes in seconds, against the exact-f32 mode of the library otherwise."""
    frames = synth.video_frames(41, n, h, w)
    loc, conf, lm = (t.cpu().numpy() for t in engine_face.face_forward(frames, MODE_F16X3))
    l32, c32, m32 = (t.cpu().numpy() for t in engine_face.face_forward(frames, MODE_FP32))
    assert np.isfinite(conf).all()
    print(h, w, "x3 vs fp32 mode: max|dconf|", np.abs(conf - c32).max(), "max|dloc|", np.abs(loc - l32).max())
    assert np.abs(conf - c32).max() < 1e-4 and np.abs(loc - l32).max() < 1e-3 and np.abs(lm - m32).max() < 1e-3
    if h * w <= 150 * 214:
        for i in range(n):
            rl, rc, rm = orf.retina_forward(sd_retina, orf.preprocess(frames[i]))
            assert np.abs(conf[i] - rc[0].numpy()).max() < 1e-4
            assert np.abs(loc[i] - rl[0].numpy()).max() < 1e-3
            assert np.abs(lm[i] - rm[0].numpy()).max() < 1e-3
    # batch invariance of the fused forms: frame 1 alone
    l1, c1, m1 = (t.cpu().numpy() for t in engine_face.face_forward(frames[1:2], MODE_F16X3))
    np.testing.assert_array_equal(c1[0], conf[1])
    np.testing.assert_array_equal(l1[0], loc[1])
    # RGB frames (the predictor's rgb=True: flipped first, retina_face_predictor.py:59-61) through the fused stem's own channel swap
    lr, cr, mr = (t.cpu().numpy() for t in engine_face.face_forward(np.ascontiguousarray(frames[..., ::-1]), MODE_F16X3, rgb=True))
    np.testing.assert_array_equal(cr, conf)
    np.testing.assert_array_equal(lr, loc)


def test_predictor_chain_matches_oracle_chain(engine_face, sd_retina):
    frame = synth.video_frames(5, 1, 120, 160)[0]
    pred = ft.RetinaFacePredictor(engine_face, sd_retina, threshold=0.5, mode=MODE_FP32)
    got = pred(frame, rgb=False)
    rl, rc, rm = orf.retina_forward(sd_retina, orf.preprocess(frame))
    ref = of.detections(rl[0].numpy(), rc[0].numpy(), rm[0].numpy(), (120, 160), threshold=0.5)
    assert got.shape == ref.shape and got.shape[0] > 0
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-2)      # pixels


def test_batch_equals_frame_by_frame(engine_face, sd_retina):
    frames = synth.video_frames(11, 4, 96, 128)
    pred = ft.RetinaFacePredictor(engine_face, sd_retina, threshold=0.3, mode=MODE_F16X3)
    together = pred.batch(frames, rgb=False)
    for t in range(4):
        np.testing.assert_array_equal(together[t], pred(frames[t], rgb=False))


@pytest.mark.parametrize("mode", [MODE_F16X3, MODE_FP32])
def test_two_lane_detector_batches_are_bit_identical(engine_face, mode):
    """Batches of at least 16 frames run as two lanes (api.hip face_forward_impl: the halves on two streams, equal passes inside a
    lane); 21 frames of 96 x 128 split 11 + 10.  Same bits as on one lane, and the outputs are complete on the caller's stream."""
    frames = synth.video_frames(21, 21, 96, 128)
    try:
        engine_face.set_static_lanes(1)
        one = [t.clone() for t in engine_face.face_forward(frames, mode)]
        engine_face.set_static_lanes(2)
        for _ in range(2):
            two = engine_face.face_forward(frames, mode)
            assert all(torch.equal(a, b) for a, b in zip(one, two))
    finally:
        engine_face.set_static_lanes(2)


def test_rejects_bad_arguments(engine_face):
    with pytest.raises(Exception):
        engine_face.face_forward(np.zeros((1, 16, 16, 3), np.uint8), MODE_FP32)


def test_other_weight_draw_against_the_oracle_in_both_parity_modes(engine):
    """The golden vectors pin seed 42; a second draw of the detector's synthetic weights against the oracle (96 x 128, two frames):
    the fused stem / chains / tails of the x3 mode and the exact-f32 mode under the same gates."""
    sd = synth.to_torch(synth.retina_state_dict(43))
    engine.load_face(sd)
    try:
        frames = synth.video_frames(8, 2, 96, 128)
        ref = [orf.retina_forward(sd, orf.preprocess(frames[i])) for i in range(2)]
        for mode in (MODE_FP32, MODE_F16X3):
            loc, conf, lm = (t.cpu().numpy() for t in engine.face_forward(frames, mode))
            for i, (rl, rc, rm) in enumerate(ref):
                assert np.abs(conf[i] - rc[0].numpy()).max() < 1e-4, mode
                assert np.abs(loc[i] - rl[0].numpy()).max() < 1e-3 and np.abs(lm[i] - rm[0].numpy()).max() < 1e-3, mode
    finally:
        engine.load_face(synth.to_torch(synth.retina_state_dict(42)))


def test_720p_frames_counting_order_and_other_tile_counts(engine_face, sd_retina):
    """1280 x 720: 37840 priors per frame -- more than the LDS sort holds, so the candidate order comes from the counting kernel --,
    stem tiles 23 x 46, layer-1 maps of 180 x 320.  The x3 mode against the exact-f32 mode of the library, and the predictor's batch
    against frame-by-frame calls (same rows, ties included)."""
    frames = synth.video_frames(5, 2, 720, 1280)
    loc, conf, lm = (t.cpu().numpy() for t in engine_face.face_forward(frames, MODE_F16X3))
    l32, c32, m32 = (t.cpu().numpy() for t in engine_face.face_forward(frames, MODE_FP32))
    assert conf.shape == (2, 37840, 2) and np.isfinite(conf).all()
    assert np.abs(conf - c32).max() < 1e-4 and np.abs(loc - l32).max() < 1e-3 and np.abs(lm - m32).max() < 1e-3
    pred = ft.RetinaFacePredictor(engine_face, sd_retina, threshold=0.3, mode=MODE_F16X3)
    together = pred.batch(frames, rgb=False)
    for t in range(2):
        one = pred(frames[t], rgb=False)
        np.testing.assert_array_equal(together[t], one)
    assert sum(len(d) for d in together) > 0
