"""Face stage (row f4) on the GPU: box decoding and crop -> tile kernels against the reference-generated vectors of
tests/golden/face.npz and the CPU oracle."""
import numpy as np
import pytest
import torch

from avcer_amd import face_tiles as ft
from avcer_amd import video_pipeline as vp
from avcer_amd.engine import MODE_FP32
from oracle import face as of
from oracle import video as ov
from test_face_cpu import G, golden_frames, golden_script

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["a", "b"])
def test_decode_all_priors(engine, name):
    size = tuple(int(v) for v in G[f"size_{name}"])
    pri = of.prior_boxes(size)
    dets = engine.face_decode(G[f"loc_{name}"], G[f"conf_{name}"], G[f"landms_{name}"], pri, size).cpu().numpy()
    h, w = size
    ref_b = of.decode(G[f"loc_{name}"], pri) * np.array([w, h, w, h], np.float32)
    ref_l = of.decode_landm(G[f"landms_{name}"], pri) * np.array([w, h] * 5, np.float32)
    np.testing.assert_allclose(dets[:, :4], ref_b, rtol=3e-6, atol=3e-5)        # expf: <= 2 ulp apart
    np.testing.assert_array_equal(dets[:, 4], G[f"conf_{name}"][:, 1])
    np.testing.assert_array_equal(dets[:, 5:], ref_l)                              # no transcendental: bit-exact


@pytest.mark.parametrize("name,thr", [("a", 0.8), ("a", 0.3), ("b", 0.8), ("b", 0.3)])
def test_detections_match_reference_predictor(engine, name, thr):
    size = tuple(int(v) for v in G[f"size_{name}"])
    got = ft.FaceDetections(engine, threshold=thr)(G[f"loc_{name}"], G[f"conf_{name}"], G[f"landms_{name}"], size)
    ref = G[f"pred_{name}_t{int(thr * 100)}"]
    assert got.shape == ref.shape and got.dtype == np.float32
    np.testing.assert_array_equal(got[:, 4], ref[:, 4])   # same scores in the same order, ties included
    # py_cpu_nms visits equal scores in whatever order numpy's default argsort leaves them (unspecified on arrays this
    # long); the kernel's rule is `argsort(kind="stable")` read backwards: ties higher prior index first.  EVERY row, tied
    # ones included, must be what that rule gives (the oracle with the stable sort), and rows whose score is unique
    # among the candidates must also be the rows of the golden file the reference itself produced.
    rule = of.detections(G[f"loc_{name}"], G[f"conf_{name}"], G[f"landms_{name}"], size, thr, stable=True)
    assert rule.shape == got.shape
    np.testing.assert_allclose(got, rule, rtol=3e-6, atol=3e-5)
    scores = G[f"conf_{name}"][:, 1]
    cand = scores[scores > 0.02]
    uniq, cnt = np.unique(cand, return_counts=True)
    tied = np.isin(got[:, 4], uniq[cnt > 1])
    np.testing.assert_allclose(got[~tied], ref[~tied], rtol=3e-6, atol=3e-5)


def test_nms_tied_scores_are_visited_higher_index_first(engine):
    """Saturated detections tie at exactly 1.0f: of two overlapping boxes with equal scores the one with the higher prior
    index is kept (`scores.argsort()[::-1]` on a tie-preserving sort), and non-overlapping tied boxes come out in that order."""
    d = np.zeros((1, 6, 15), np.float32)
    d[0, :, :5] = [[10, 10, 50, 50, 1.0], [12, 12, 52, 52, 1.0],      # overlap: index 1 wins
                   [200, 200, 240, 240, 0.5], [201, 201, 241, 241, 0.5],  # overlap: index 3 wins
                   [400, 10, 440, 50, 1.0], [300, 300, 340, 340, 0.9]]
    d[0, :, 5] = np.arange(6)
    out, cnt = engine.face_nms(d, conf_thresh=0.02, nms_thresh=0.4, nms_top_k=5000, top_k=750, threshold=0.3)
    got = out[0, :int(cnt[0]), 5].cpu().numpy().astype(int).tolist()
    assert got == [4, 1, 5, 3]
    assert got == [int(i) for i in of.nms(d[0, :, :5], 0.4, 5000, stable=True)]


def test_nms_kernel_matches_reference_keep_lists(engine):
    """avcer_face_nms against the keep lists py_cpu_nms itself produced (face.npz), a random dense case against the
    oracle, and a batch of frames in one call."""
    d = G["nms_dets"]

    def run(dets5, thresh, nms_top_k, top_k=1024):
        rows = np.zeros((1, len(dets5), 15), np.float32)
        rows[0, :, :5] = dets5
        rows[0, :, 5] = np.arange(len(dets5))          # carries the row index through the kernel
        out, cnt = engine.face_nms(rows, conf_thresh=-1e30, nms_thresh=thresh, nms_top_k=nms_top_k, top_k=top_k, threshold=-1e30)
        return out[0, :int(cnt[0]), 5].cpu().numpy().astype(int).tolist()

    assert run(d, 0.4, 5000) == list(G["nms_keep_04"])
    assert run(d, 0.2, 50) == list(G["nms_keep_02_top50"])
    rng = np.random.default_rng(5)
    n = 3000
    xy = rng.uniform(0, 600, (n, 2)).astype(np.float32)
    wh = rng.uniform(5, 120, (n, 2)).astype(np.float32)
    dense = np.concatenate([xy, xy + wh, rng.uniform(0, 1, (n, 1)).astype(np.float32)], axis=1)
    assert run(dense, 0.4, 5000) == [int(i) for i in of.nms(dense, 0.4, 5000)][:1024]
    assert run(dense, 0.3, 700, top_k=100) == [int(i) for i in of.nms(dense, 0.3, 700)][:100]
    # two frames with different candidate counts in one launch; confidence floor and final threshold applied
    batch = np.zeros((2, n, 15), np.float32)
    batch[0, :, :5], batch[1, :, :5] = dense, dense[::-1]
    batch[1, n // 2:, 4] = 0.0                        # below the floor: not candidates
    batch[:, :, 5] = np.arange(n)
    out, cnt = engine.face_nms(batch, conf_thresh=0.02, nms_thresh=0.4, nms_top_k=5000, top_k=750, threshold=0.8)
    for f in range(2):
        b = batch[f]
        inds = np.where(b[:, 4] > 0.02)[0]
        keep = inds[of.nms(b[inds, :5], 0.4, 5000)][:750]
        keep = keep[b[keep, 4] >= 0.8]
        assert out[f, :int(cnt[f]), 5].cpu().numpy().astype(int).tolist() == keep.tolist()


def test_nms_order_sort_and_counting_forms_agree(engine):
    """The descending-score order comes from an LDS bitonic sort for frames of at most 16384 priors and from the counting kernel
    beyond (1280 x 720: 37840 priors).  Same rule (ties: higher prior index first) -- checked on one score table with heavy ties,
    negative scores and non-candidates, once as 16000 priors (sort) and once padded to 20000 (counting) with non-candidates."""
    rng = np.random.default_rng(11)
    n = 16000
    xy = rng.uniform(0, 3000, (n, 2)).astype(np.float32)
    wh = rng.uniform(5, 60, (n, 2)).astype(np.float32)
    score = np.round(rng.uniform(-1.0, 0.2, n), 3).astype(np.float32)          # ~1300 candidates on ~100 distinct scores: ties everywhere
    d = np.zeros((1, n, 15), np.float32)
    d[0, :, :4], d[0, :, 4], d[0, :, 5] = np.concatenate([xy, xy + wh], 1), score, np.arange(n)
    pad = np.zeros((1, 20000, 15), np.float32)
    pad[0, :n] = d[0]
    pad[0, n:, 4] = -5.0                                                       # below the floor
    res = []
    for rows in (d, pad):
        out, cnt = engine.face_nms(rows, conf_thresh=0.1, nms_thresh=0.4, nms_top_k=5000, top_k=750, threshold=0.15)
        res.append(out[0, :int(cnt[0]), 5].cpu().numpy().astype(int).tolist())
    inds = np.where(score > 0.1)[0]
    keep = inds[of.nms(d[0, inds, :5], 0.4, 5000, stable=True)][:750]
    keep = keep[score[keep] >= 0.15]
    assert res[0] == res[1] == keep.tolist() and len(keep) > 100


@pytest.mark.parametrize("n", [6000, 12000])
def test_nms_order_with_thousands_of_candidates(engine, n):
    """The candidate order sorts the next power of two of the candidate COUNT (2048 ... 16384 keys: four instantiations of the
    unrolled bitonic passes); here 6000 and 12000 candidates -- the 8192- and 16384-key sorts, the bench's synthetic detector is the
    latter -- on clustered boxes with tied scores, so that the walk keeps few boxes and the plain-loop oracle finishes in seconds.
    The kept list is a function of the whole order: a single misplaced key shows."""
    rng = np.random.default_rng(n)
    centres = rng.uniform(50, 1500, (60, 2)).astype(np.float32)
    c = centres[rng.integers(0, 60, n)] + rng.normal(0, 3, (n, 2)).astype(np.float32)
    wh = rng.uniform(60, 90, (n, 2)).astype(np.float32)
    score = np.round(rng.uniform(0.03, 1.0, n), 3).astype(np.float32)           # ~970 distinct values: ties everywhere
    d = np.zeros((1, n, 15), np.float32)
    d[0, :, :4], d[0, :, 4], d[0, :, 5] = np.concatenate([c - wh / 2, c + wh / 2], 1), score, np.arange(n)
    out, cnt = engine.face_nms(d, conf_thresh=0.02, nms_thresh=0.4, nms_top_k=5000, top_k=750, threshold=0.05)
    got = out[0, :int(cnt[0]), 5].cpu().numpy().astype(int).tolist()
    inds = np.where(score > 0.02)[0]
    keep = inds[of.nms(d[0, inds, :5], 0.4, 5000, stable=True)][:750]          # the 5000 best of them, ties higher index first
    keep = keep[score[keep] >= 0.05]
    assert len(inds) == n and 30 < len(keep) < 750 and got == keep.tolist()


def test_decode_batch_equals_per_frame_decode(engine):
    size = tuple(int(v) for v in G["size_a"])
    pri = of.prior_boxes(size)
    rng = np.random.default_rng(2)
    loc = np.stack([G["loc_a"], G["loc_a"][::-1].copy(), rng.normal(0, 1, G["loc_a"].shape).astype(np.float32)])
    conf = np.stack([G["conf_a"], G["conf_a"][::-1].copy(), G["conf_a"]])
    lm = np.stack([G["landms_a"], G["landms_a"][::-1].copy(), rng.normal(0, 1, G["landms_a"].shape).astype(np.float32)])
    got = engine.face_decode_batch(loc, conf, lm, pri, size).cpu()
    for t in range(3):
        assert torch.equal(got[t], engine.face_decode(loc[t], conf[t], lm[t], pri, size).cpu())


def test_detections_empty(engine):
    p = len(G["priors_a"])
    conf = np.tile(np.array([[1.0, 0.0]], dtype=np.float32), (p, 1))
    got = ft.FaceDetections(engine)(np.zeros((p, 4), np.float32), conf, np.zeros((p, 10), np.float32), (120, 160))
    assert got.shape == (0, 15) and got.dtype == np.float32


def test_video_tiler_matches_reference_process(engine):
    records, tiles = ft.VideoTiler(engine).process(golden_frames(), golden_script())
    np.testing.assert_array_equal(records, G["track_records"])
    assert tiles.dtype == torch.uint8 and tiles.is_cuda
    np.testing.assert_array_equal(tiles.cpu().numpy(), G["track_tiles"])           # bit-exact u8, BGR -> RGB + NEAREST


def test_crop_tiles_rgb_identity_and_bad_rects(engine):
    fr = np.random.default_rng(3).integers(0, 256, (2, 300, 260, 3), dtype=np.uint8)
    rects = np.array([[0, 10, 20, 234, 244],      # 224x224 crop: plain copy
                      [1, 0, 0, 260, 300],        # whole frame, down-sampled
                      [1, 5, 7, 6, 8],            # 1x1 crop replicated
                      [0, 100, 100, 100, 150],    # empty -> zeros
                      [2, 0, 0, 10, 10],          # frame out of range -> zeros
                      [0, 200, 0, 270, 50]],      # leaves the frame -> zeros
                     dtype=np.int32)
    t = engine.crop_tiles(fr, rects, bgr=False).cpu().numpy()
    np.testing.assert_array_equal(t[0], fr[0, 20:244, 10:234])
    np.testing.assert_array_equal(t[1], ov.nearest_resize_u8(fr[1]))
    assert (t[2] == fr[1, 7, 5]).all()
    assert not t[3].any() and not t[4].any() and not t[5].any()
    tb = engine.crop_tiles(fr, rects[:2], bgr=True).cpu().numpy()
    np.testing.assert_array_equal(tb, t[:2, :, :, ::-1])


def test_track_feeds_visual_path(engine_static, sd_static, sd_dynamic):
    """tiles of one track -> the visual path, against the oracle run on the reference-generated tiles."""
    engine_static.load_dynamic(sd_dynamic)
    records, tiles = ft.VideoTiler(engine_static).process(golden_frames(), golden_script())
    total = int(G["track_T"])
    frames, present = ft.track_clip(records, tiles, 1, total)          # face B's first visit: frames 3..7
    assert present.tolist() == [3 <= t < 8 for t in range(total)]
    stat, dyn = vp.visual_forward(engine_static, frames, present, 25, MODE_FP32)
    rows = np.where(G["track_records"][:, 1] == 1)[0]
    ref_frames = np.zeros((total, 224, 224, 3), np.uint8)
    ref_frames[G["track_records"][rows, 0]] = G["track_tiles"][rows]
    rs, rd = ov.visual_forward(sd_static, sd_dynamic, ref_frames, present, 25, batched=True)
    np.testing.assert_allclose(stat.cpu().numpy(), rs, rtol=0, atol=1e-4)
    print("track -> visual path: max|dprob|", np.abs(stat.cpu().numpy() - rs).max(), "max|d dynamic logit|", np.abs(dyn.cpu().numpy() - rd).max())
    np.testing.assert_allclose(dyn.cpu().numpy(), rd, rtol=0, atol=1e-5)  # measured 6.6e-7
