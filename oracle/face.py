"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the face stage around the RetinaFace network.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product path
(avcer_amd/) never does.  Pinned by tests/golden/face.npz, which was produced by running the reference's own
PriorBox, RetinaFacePredictor.__call__, py_cpu_nms, SimpleFaceTracker and VideoPredictor.process around a stand-in
network (tests/golden/make_golden.py gen_face).  The network itself (torchvision ResNet-50 + FPN + SSH) is NOT restated:
torchvision and the weights are absent, so it could not be pinned.

Plain loops on purpose: this is the checker, not the thing measured.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.optimize import linear_sum_assignment

from .video import nearest_resize_u8

# retina_face/config.py:22-39 (cfg_re50) and retina_face_predictor.py:52-55 (create_config)
MIN_SIZES = ((16, 32), (64, 128), (256, 512))
STEPS = (8, 16, 32)
VARIANCE = (0.1, 0.2)
TOP_K, CONF_THRESH, NMS_THRESH, NMS_TOP_K = 750, 0.02, 0.4, 5000


def prior_boxes(image_size) -> np.ndarray:
    """prior_box.py:16-33: (cx, cy, w, h) in image fractions, level-major, row-major cells, two sizes per cell."""
    h, w = image_size
    rows = []
    for k, step in enumerate(STEPS):
        fh, fw = math.ceil(h / step), math.ceil(w / step)
        for i in range(fh):
            for j in range(fw):
                for ms in MIN_SIZES[k]:
                    rows.append([(j + 0.5) * step / w, (i + 0.5) * step / h, ms / w, ms / h])
    return np.array(rows, dtype=np.float64).astype(np.float32)


def decode(loc, priors, variance=VARIANCE) -> np.ndarray:
    """box_utils.py:210-228 in float32: centre/size -> corners."""
    v0, v1 = np.float32(variance[0]), np.float32(variance[1])
    out = np.empty((len(loc), 4), dtype=np.float32)
    for i in range(len(loc)):
        cx = priors[i, 0] + loc[i, 0] * v0 * priors[i, 2]
        cy = priors[i, 1] + loc[i, 1] * v0 * priors[i, 3]
        w = priors[i, 2] * np.exp(loc[i, 2] * v1)
        h = priors[i, 3] * np.exp(loc[i, 3] * v1)
        x0 = cx - w / np.float32(2)
        y0 = cy - h / np.float32(2)
        out[i] = (x0, y0, w + x0, h + y0)
    return out


def decode_landm(pre, priors, variance=VARIANCE) -> np.ndarray:
    """box_utils.py:231-249."""
    v0 = np.float32(variance[0])
    out = np.empty((len(pre), 10), dtype=np.float32)
    for k in range(5):
        out[:, 2 * k] = priors[:, 0] + pre[:, 2 * k] * v0 * priors[:, 2]
        out[:, 2 * k + 1] = priors[:, 1] + pre[:, 2 * k + 1] * v0 * priors[:, 3]
    return out


def nms(dets, thresh, top_k, stable=False):
    """py_cpu_nms.py:11-39 (areas with the +1 pixel convention, float32).  `stable`: the same expression with
    `argsort(kind="stable")`, which pins the order of exactly tied scores (higher index first after the reversal);
    numpy's default sort leaves it unspecified on long arrays."""
    x1, y1, x2, y2, scores = (dets[:, c] for c in range(5))
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = list((scores.argsort(kind="stable") if stable else scores.argsort())[: -top_k - 1: -1])
    keep = []
    while order:
        i = order.pop(0)
        keep.append(int(i))
        rest = []
        for j in order:
            w = max(np.float32(0.0), min(x2[i], x2[j]) - max(x1[i], x1[j]) + 1)
            h = max(np.float32(0.0), min(y2[i], y2[j]) - max(y1[i], y1[j]) + 1)
            inter = np.float32(w * h)
            if inter / (areas[i] + areas[j] - inter) <= np.float32(thresh):
                rest.append(j)
        order = rest
    return keep


def detections(loc, conf, landms, image_size, threshold=0.8, stable=False) -> np.ndarray:
    """retina_face_predictor.py:58-108 after `self.net(image)`: decode, scale, confidence floor, NMS, top-k, threshold."""
    h, w = image_size
    priors = prior_boxes(image_size)
    boxes = decode(loc, priors) * np.array([w, h, w, h], dtype=np.float32)
    lm = decode_landm(landms, priors) * np.array([w, h] * 5, dtype=np.float32)
    scores = conf[:, 1]
    inds = np.where(scores > CONF_THRESH)[0]
    if len(inds) == 0:
        return np.empty((0, 15), dtype=np.float32)
    dets = np.hstack((boxes[inds], scores[inds, None])).astype(np.float32)
    lm = lm[inds]
    keep = nms(dets, NMS_THRESH, NMS_TOP_K, stable)
    dets = np.concatenate((dets[keep][:TOP_K], lm[keep][:TOP_K]), axis=1)
    sel = np.where(dets[:, 4] >= threshold)[0]
    return dets[sel] if len(sel) else np.empty((0, 15), dtype=np.float32)


class Tracker:
    """utils/simple_face_tracker.py:10-85 (IoU distance + Hungarian assignment, ids from 1)."""

    def __init__(self, iou_threshold=0.4, minimum_face_size=0.0):
        self.thr = float(np.clip(1.0 - iou_threshold, 0.0, 1.0))
        self.min_area = max(minimum_face_size ** 2, np.finfo(float).eps)
        self.tracklets = []
        self.counter = 0

    def __call__(self, boxes):
        if boxes.size <= 0:
            self.tracklets = []
            return []
        areas = np.abs((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
        for t in self.tracklets:
            t["tracked"] = False
        dist = np.full((len(boxes), len(self.tracklets)), 2.0 * min(len(boxes), len(self.tracklets)), dtype=float)
        for r, fb in enumerate(boxes):
            if areas[r] < self.min_area:
                continue
            for c, t in enumerate(self.tracklets):
                tb = t["bbox"]
                xl = max(min(fb[0], fb[2]), min(tb[0], tb[2]))
                yt = max(min(fb[1], fb[3]), min(tb[1], tb[3]))
                xr = min(max(fb[2], fb[0]), max(tb[2], tb[0]))
                yb = min(max(fb[3], fb[1]), max(tb[3], tb[1]))
                if xr <= xl or yb <= yt:
                    d = 1.0
                else:
                    inter = (xr - xl) * (yb - yt)
                    d = 1.0 - inter / float(areas[r] + t["area"] - inter)
                if d <= self.thr:
                    dist[r, c] = d
        ids = [None] * len(boxes)
        for r, c in zip(*linear_sum_assignment(dist)):
            if dist[r, c] <= self.thr:
                ids[r] = self.tracklets[c]["id"]
                self.tracklets[c].update(bbox=boxes[r, :4].copy(), area=areas[r], tracked=True)
        self.tracklets = [t for t in self.tracklets if t["tracked"]]
        for r, fb in enumerate(boxes):
            if areas[r] >= self.min_area and ids[r] is None:
                self.counter += 1
                self.tracklets.append({"bbox": fb[:4].copy(), "area": areas[r], "id": self.counter, "tracked": True})
                ids[r] = self.counter
        return ids


def crop_rect(det, w, h):
    """get_face_images.py:52-56: truncate to int, clamp, then numpy's slice rules for fr[y0:y1, x0:x1]."""
    sx, sy, ex, ey = (int(v) for v in det[:4].astype(int))
    sx, sy = max(0, sx), max(0, sy)
    ex, ey = min(w - 1, ex), min(h - 1, ey)
    x0, x1, _ = slice(sx, ex).indices(w)
    y0, y1, _ = slice(sy, ey).indices(h)
    return x0, y0, max(x1, x0), max(y1, y0)


def process_video(frames_bgr, dets_per_frame):
    """get_face_images.py:38-63 without the JPEG files: -> records (frame, track dir, x0, y0, x1, y1) and RGB tiles."""
    h, w = frames_bgr.shape[1:3]
    tracker = Tracker(0.4, 0.0)
    recs, tiles = [], []
    for t, dets in enumerate(dets_per_frame):
        ids = tracker(dets)
        for det, tid in zip(dets, ids):
            x0, y0, x1, y1 = crop_rect(det, w, h)
            recs.append((t, tid - 1, x0, y0, x1, y1))
            crop = frames_bgr[t, y0:y1, x0:x1]
            tiles.append(nearest_resize_u8(np.ascontiguousarray(crop[..., ::-1])))
    return np.array(recs, dtype=np.int64).reshape(-1, 6), np.stack(tiles) if tiles else np.zeros((0, 224, 224, 3), np.uint8)
