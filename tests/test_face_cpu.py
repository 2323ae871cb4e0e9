"""Face stage (row f4), CPU part: the oracle and the host-side mirror against vectors produced by the reference's own
PriorBox / RetinaFacePredictor.__call__ / py_cpu_nms / SimpleFaceTracker / VideoPredictor.process
(tests/golden/make_golden.py gen_face)."""
import os

import numpy as np
import pytest

from avcer_amd import face_tiles as ft
from oracle import face as of

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "face.npz"))


def golden_frames():
    h, w = (int(v) for v in G["track_hw"])
    t_total = int(G["track_T"])
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([np.stack([x & 255, y & 255, (3 * x + 5 * y + 17 * t) & 255], axis=-1).astype(np.uint8)
                     for t in range(t_total)])


def golden_script():
    return [G[f"track_dets_{t}"] for t in range(int(G["track_T"]))]


@pytest.mark.parametrize("name", ["a", "b"])
def test_prior_boxes(name):
    size = tuple(int(v) for v in G[f"size_{name}"])
    ref = G[f"priors_{name}"]
    for fn in (of.prior_boxes, ft.prior_boxes):
        got = fn(size)
        assert got.dtype == np.float32 and got.shape == ref.shape
        np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("name,thr", [("a", 0.8), ("a", 0.3), ("b", 0.8), ("b", 0.3)])
def test_oracle_detections(name, thr):
    size = tuple(int(v) for v in G[f"size_{name}"])
    got = of.detections(G[f"loc_{name}"], G[f"conf_{name}"], G[f"landms_{name}"], size, threshold=thr)
    ref = G[f"pred_{name}_t{int(thr * 100)}"]
    assert got.shape == ref.shape and got.dtype == np.float32
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-5)   # torch vs numpy expf


def test_oracle_detections_empty():
    p = len(G["priors_a"])
    conf = np.tile(np.array([[1.0, 0.0]], dtype=np.float32), (p, 1))
    got = of.detections(np.zeros((p, 4), np.float32), conf, np.zeros((p, 10), np.float32), (120, 160))
    assert got.shape == (0, 15) == G["pred_empty"].shape


@pytest.mark.parametrize("fn", [of.nms])  # the product path runs NMS on the GPU: tests/test_gpu_face.py
def test_nms(fn):
    assert list(fn(G["nms_dets"], 0.4, 5000)) == list(G["nms_keep_04"])
    assert list(fn(G["nms_dets"], 0.2, 50)) == list(G["nms_keep_02_top50"])


def test_tracker_and_rects_match_reference_process():
    h, w = (int(v) for v in G["track_hw"])
    ref = G["track_records"]
    # oracle: full process incl. tiles
    recs, tiles = of.process_video(golden_frames(), golden_script())
    np.testing.assert_array_equal(recs, ref)
    np.testing.assert_array_equal(tiles, G["track_tiles"])
    # product host logic (tracker + rects), no GPU
    tr = ft.SimpleFaceTracker(0.4, 0.0)
    got = []
    for t, dets in enumerate(golden_script()):
        for rect, tid in zip(ft.crop_rects(dets, w, h), tr(dets)):
            got.append((t, tid - 1, *rect))
    np.testing.assert_array_equal(np.array(got, dtype=np.int64), ref)
    assert sorted(set(ref[:, 1])) == [0, 1, 2, 3, 4, 5]     # jump, leave/return and the empty frame each start a new id


def test_tracker_edge_cases():
    tr = ft.SimpleFaceTracker(0.4, 0.0)
    assert tr(np.zeros((0, 15), np.float32)) == []
    a = np.array([[10, 10, 50, 50, 0.9], [12, 11, 52, 51, 0.8]], dtype=np.float32)     # two faces on one tracklet later
    assert tr(a) == [1, 2]
    assert tr(a[::-1].copy()) == [2, 1]                      # assignment follows the boxes, not the row order
    z = np.array([[5, 5, 5, 40, 0.9]], dtype=np.float32)     # zero area: no id (the reference then fails on tid - 1)
    assert tr(z) == [None]
    tr.reset()
    assert tr(a) == [1, 2]
    ref_tr, mine = of.Tracker(0.4, 0.0), ft.SimpleFaceTracker(0.4, 0.0)
    rng = np.random.default_rng(5)
    for _ in range(40):                                       # random walks incl. births and deaths
        k = int(rng.integers(0, 5))
        c = rng.uniform(20, 200, (k, 2)); s = rng.uniform(10, 60, (k, 2))
        b = np.concatenate([c - s / 2, c + s / 2, rng.uniform(0.8, 1, (k, 1))], 1).astype(np.float32)
        assert ref_tr(b) == mine(b)


def test_crop_rects_slice_rules():
    d = np.array([[-3.7, -0.2, 20.9, 30.1], [150.2, 100.9, 400.0, 400.0], [-50.0, 10.0, -5.0, 40.0]], dtype=np.float32)
    got = ft.crop_rects(d, 160, 120)
    assert got[0].tolist() == [0, 0, 20, 30]                 # truncation toward zero, start clamped to 0
    assert got[1].tolist() == [150, 100, 159, 119]           # end clamped to size-1 and exclusive
    assert got[2].tolist() == list(of.crop_rect(d[2], 160, 120))   # negative end: numpy slice wrap, as fr[..., 0:-5]
    for k in range(3):
        assert got[k].tolist() == list(of.crop_rect(d[k], 160, 120))
