"""Face stage around the RetinaFace network ("next" row f4): detections -> tracks -> u8 tiles.

Mirrors, with the reference's names and argument meaning:
  * `PriorBox(cfg_re50, image_size).forward()`            retina_face/prior_box.py:16-33        -> prior_boxes
  * `RetinaFacePredictor.__call__` after `self.net(image)`  retina_face_predictor.py:70-108      -> FaceDetections
  * `py_cpu_nms`                                          retina_face/py_cpu_nms.py:11-39        -> nms
  * `SimpleFaceTracker`                                   utils/simple_face_tracker.py:10-90     -> SimpleFaceTracker
  * `VideoPredictor.process`                              data/get_face_images.py:38-63          -> VideoTiler.process

What runs where: box / landmark decoding, the confidence filter + NMS + top-k + final threshold (one launch per batch
of frames, only the kept rows return to the host) and the crop + NEAREST resize into the u8 tile buffer are HIP kernels
(`avcer_face_decode`, `avcer_face_nms`, `avcer_crop_tiles`); the IoU/Hungarian tracker acts on a handful of boxes per
frame, is sequential in time and stays on the host, as in the reference (scipy there too).  The RetinaFace-R50
network runs on the same implicit-GEMM kernel as the recognition models (`avcer_face_forward`, mirror
`RetinaFacePredictor` below).  Tiles go straight to `avcer_static_forward`; the reference's JPEG file round trip
(cv2.imwrite -> PIL.Image.open) is gone, which is the only intended difference.  Video decoding is not part of this build.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import List, Optional, Sequence

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from .engine import MODE_DEFAULT

# retina_face/config.py:22-39 (cfg_re50)
CFG_RE50 = {"min_sizes": [[16, 32], [64, 128], [256, 512]], "steps": [8, 16, 32], "variance": [0.1, 0.2], "clip": False}


@lru_cache(maxsize=8)
def _prior_boxes(h: int, w: int) -> np.ndarray:
    levels = []
    for sizes, step in zip(CFG_RE50["min_sizes"], CFG_RE50["steps"]):
        fh, fw = math.ceil(h / step), math.ceil(w / step)
        cy = ((np.arange(fh, dtype=np.float64) + 0.5) * step / h)[:, None, None]
        cx = ((np.arange(fw, dtype=np.float64) + 0.5) * step / w)[None, :, None]
        sz = np.asarray(sizes, dtype=np.float64)[None, None, :]
        lvl = np.stack(np.broadcast_arrays(cx, cy, sz / w, sz / h), axis=-1)  # [fh, fw, sizes, 4]
        levels.append(lvl.reshape(-1, 4))
    out = np.concatenate(levels).astype(np.float32)
    if CFG_RE50["clip"]:
        out = np.clip(out, 0.0, 1.0)
    out.setflags(write=False)
    return out


def prior_boxes(image_size) -> np.ndarray:
    """Anchors (cx, cy, w, h) as fractions of the image for an image of (height, width)."""
    return _prior_boxes(int(image_size[0]), int(image_size[1]))


class FaceDetections:
    """The post-network half of `RetinaFacePredictor`: same constructor thresholds, same [k,15] float32 result."""

    def __init__(self, engine, threshold: float = 0.8, top_k: int = 750, conf_thresh: float = 0.02,
                 nms_thresh: float = 0.4, nms_top_k: int = 5000):
        self.engine = engine
        self.threshold = threshold
        self.top_k, self.conf_thresh, self.nms_thresh, self.nms_top_k = top_k, conf_thresh, nms_thresh, nms_top_k
        self._priors_dev = {}

    def __call__(self, loc, conf, landms, image_size) -> np.ndarray:
        """loc [P,4], conf [P,2] (softmaxed), landms [P,10] as the network's test phase returns them (batch dim
        squeezed), for an image of (height, width)."""
        key = (int(image_size[0]), int(image_size[1]))
        if key not in self._priors_dev:
            self._priors_dev[key] = torch.from_numpy(np.array(prior_boxes(key))).to(self.engine.device)
        dets = self.engine.face_decode(loc, conf, landms, self._priors_dev[key], key, CFG_RE50["variance"])
        # confidence floor, NMS (py_cpu_nms order), top-k and the final threshold run on the device: only the kept rows
        # and their count come back to the host
        rows, cnt = self.engine.face_nms(dets[None], self.conf_thresh, self.nms_thresh, self.nms_top_k, self.top_k, self.threshold)
        k = int(cnt[0])
        return rows[0, :k].cpu().numpy() if k else np.empty((0, 15), dtype=np.float32)


    def batch(self, loc, conf, landms, image_size) -> List[np.ndarray]:
        """[T,P,*] network outputs of T frames -> one [k,15] array per frame: ONE decode launch, ONE order + NMS launch pair and
        ONE device-to-host copy for the whole batch."""
        key = (int(image_size[0]), int(image_size[1]))
        if key not in self._priors_dev:
            self._priors_dev[key] = torch.from_numpy(np.array(prior_boxes(key))).to(self.engine.device)
        dets = self.engine.face_decode_batch(loc, conf, landms, self._priors_dev[key], key, CFG_RE50["variance"])
        rows, cnt = self.engine.face_nms(dets, self.conf_thresh, self.nms_thresh, self.nms_top_k, self.top_k, self.threshold)
        cnt = cnt.cpu().numpy()
        most = int(cnt.max()) if len(cnt) else 0
        if not most:
            return [np.empty((0, 15), dtype=np.float32) for _ in range(len(cnt))]
        # only the rows some frame kept cross to the host, through a page-locked buffer (torch's host allocator caches the block from
        # call to call; a pageable copy of the worst case -- 750 frames x 750 rows, 34 MB -- is 5 ms with the GPU idle behind it,
        # pinned ~1.4 ms).  The per-frame arrays are VIEWS of that buffer, which they keep alive: no second copy on the host.
        host = torch.empty((int(rows.shape[0]), most, 15), dtype=torch.float32, pin_memory=True)
        host.copy_(rows[:, :most], non_blocking=True)
        torch.cuda.current_stream(rows.device).synchronize()
        rows = host.numpy()
        return [rows[t, :int(cnt[t])] if cnt[t] else np.empty((0, 15), dtype=np.float32) for t in range(len(cnt))]


class RetinaFacePredictor:
    """`RetinaFacePredictor(threshold, device, model)` (retina_face_predictor.py:17-108) on the HIP path: the network,
    box decoding and the device-side filter / NMS / top-k (`FaceDetections`).  `state_dict` = RetinaFace(cfg_re50).state_dict() (the file
    `Resnet50_Final.pth`, with or without the `module.` prefix)."""

    def __init__(self, engine, state_dict, threshold: float = 0.8, mode: int = MODE_DEFAULT):
        self.engine, self.mode = engine, mode
        engine.load_face(state_dict)
        self.post = FaceDetections(engine, threshold=threshold)

    def __call__(self, image, rgb: bool = True) -> np.ndarray:
        """image u8 [H,W,3] -> detections [k,15] = x0, y0, x1, y1, score, 5 landmarks (float32)."""
        img = image if torch.is_tensor(image) else torch.from_numpy(np.ascontiguousarray(image))
        loc, conf, lm = self.engine.face_forward(img[None], self.mode, rgb=rgb)
        return self.post(loc[0], conf[0], lm[0], (int(img.shape[0]), int(img.shape[1])))

    def batch(self, frames, rgb: bool = False) -> List[np.ndarray]:
        """frames u8 [T,H,W,3] -> one detection array per frame; the network runs once over the whole batch (the
        reference calls it frame by frame, get_face_images.py:49)."""
        x = frames if torch.is_tensor(frames) else torch.from_numpy(np.ascontiguousarray(frames))
        loc, conf, lm = self.engine.face_forward(x, self.mode, rgb=rgb)
        size = (int(x.shape[1]), int(x.shape[2]))
        return self.post.batch(loc, conf, lm, size)


class SimpleFaceTracker:
    """IoU + Hungarian face tracker; ids start at 1, a frame without faces drops every tracklet."""

    def __init__(self, iou_threshold: float = 0.4, minimum_face_size: float = 0.0) -> None:
        self.iou_threshold = iou_threshold
        self.minimum_face_size = minimum_face_size
        self.reset()

    def reset(self, reset_tracklet_counter: bool = True) -> None:
        self._boxes = np.zeros((0, 4), dtype=np.float32)
        self._areas = np.zeros((0,), dtype=np.float32)
        self._ids: List[int] = []
        if reset_tracklet_counter:
            self._counter = 0

    def __call__(self, face_boxes: np.ndarray) -> List[Optional[int]]:
        if face_boxes.size <= 0:
            self.reset(False)
            return []
        fb = face_boxes[:, :4]
        areas = np.abs((fb[:, 2] - fb[:, 0]) * (fb[:, 3] - fb[:, 1]))
        thr = float(np.clip(1.0 - self.iou_threshold, 0.0, 1.0))
        big = areas >= max(self.minimum_face_size ** 2, np.finfo(float).eps)
        n, m = len(fb), len(self._ids)
        dist = np.full((n, m), 2.0 * min(n, m), dtype=float)
        if m:
            tb = self._boxes
            xl = np.maximum(np.minimum(fb[:, 0], fb[:, 2])[:, None], np.minimum(tb[:, 0], tb[:, 2])[None])
            yt = np.maximum(np.minimum(fb[:, 1], fb[:, 3])[:, None], np.minimum(tb[:, 1], tb[:, 3])[None])
            xr = np.minimum(np.maximum(fb[:, 0], fb[:, 2])[:, None], np.maximum(tb[:, 0], tb[:, 2])[None])
            yb = np.minimum(np.maximum(fb[:, 1], fb[:, 3])[:, None], np.maximum(tb[:, 1], tb[:, 3])[None])
            inter = (xr - xl) * (yb - yt)                                    # dtype of the boxes, like the scalars there
            union = (areas[:, None] + self._areas[None]) - inter
            # evaluated in the boxes' dtype: `1.0 - f32 / float(f32)` is float32 under NumPy >= 2 (weak Python scalars);
            # NumPy 1.x made the last two operations float64, a < 1e-7 difference that only matters for exact ties
            with np.errstate(divide="ignore", invalid="ignore"):
                d = (1.0 - inter / union).astype(float)
            d = np.where((xr <= xl) | (yb <= yt), 1.0, d)
            ok = (d <= thr) & big[:, None]
            dist[ok] = d[ok]
        ids: List[Optional[int]] = [None] * n
        tracked = np.zeros(m, dtype=bool)
        boxes, tareas = self._boxes.copy(), self._areas.copy()
        for r, c in zip(*linear_sum_assignment(dist)):
            if dist[r, c] <= thr:
                ids[r] = self._ids[c]
                boxes[c], tareas[c], tracked[c] = fb[r], areas[r], True
        new = [r for r in range(n) if big[r] and ids[r] is None]
        for r in new:
            self._counter += 1
            ids[r] = self._counter
        self._boxes = np.concatenate([boxes[tracked], fb[new].astype(boxes.dtype)]) if m or new else boxes
        self._areas = np.concatenate([tareas[tracked], areas[new].astype(tareas.dtype)])
        self._ids = [i for i, t in zip(self._ids, tracked) if t] + [ids[r] for r in new]
        return ids


def crop_rects(dets: np.ndarray, w: int, h: int) -> np.ndarray:
    """Box corners -> the half-open pixel rectangle `fr[y0:y1, x0:x1]` the reference crops (truncate toward zero, clamp
    the start to 0 and the end to size-1, then numpy's slice rules).  [k,>=4] float -> [k,4] int (x0, y0, x1, y1)."""
    b = np.asarray(dets)[:, :4].astype(int)
    out = np.empty((len(b), 4), dtype=np.int64)
    for k, (sx, sy, ex, ey) in enumerate(b):
        x0, x1, _ = slice(max(0, int(sx)), min(w - 1, int(ex))).indices(w)
        y0, y1, _ = slice(max(0, int(sy)), min(h - 1, int(ey))).indices(h)
        out[k] = (x0, y0, max(x0, x1), max(y0, y1))
    return out


class VideoTiler:
    """`VideoPredictor.process` with the decoded frames and the per-frame detections already in memory: tracks the
    faces and cuts every detection into a 224x224 RGB tile on the GPU.  Returns per write, in the reference's order:
    records int64 [n,6] = frame index, track directory (tid-1), x0, y0, x1, y1; and the tiles u8 [n,224,224,3] (device)."""

    def __init__(self, engine):
        self.engine = engine
        self.face_tracker = SimpleFaceTracker(iou_threshold=0.4, minimum_face_size=0.0)

    def process(self, frames_bgr, dets_per_frame: Sequence[np.ndarray]):
        frames = frames_bgr if torch.is_tensor(frames_bgr) else torch.from_numpy(np.ascontiguousarray(frames_bgr))
        if frames.dim() != 4 or frames.shape[-1] != 3 or frames.dtype != torch.uint8:
            raise ValueError("frames must be uint8 [T,H,W,3] (BGR, as cv2 decodes them)")
        t_total, h, w = (int(v) for v in frames.shape[:3])
        if len(dets_per_frame) != t_total:
            raise ValueError("one detection array per frame")
        # tracker + crop rectangles of the whole video in ONE native host call (csrc/track.hip: the arithmetic of SimpleFaceTracker
        # above and of crop_rects, scipy's assignment algorithm): the 750 Python iterations this replaces took 25-30 ms per 30 s
        # video, with the visual branch's stream idle behind them
        records = self.engine.track_faces(dets_per_frame, w, h, self.face_tracker.iou_threshold, self.face_tracker.minimum_face_size)
        if not len(records):
            return records, torch.zeros((0, 224, 224, 3), dtype=torch.uint8, device=self.engine.device)
        rects = torch.from_numpy(records[:, [0, 2, 3, 4, 5]].astype(np.int32))
        return records, self.engine.crop_tiles(frames, rects, bgr=True)


def track_clip(records: np.ndarray, tiles: torch.Tensor, track: int, total_frames: int):
    """One track directory as `video_pipeline.visual_forward(frames, present, fps)` consumes it: frames u8
    [total_frames,224,224,3] with the track's tiles at their frame indices (other frames zero, never read) and the
    present mask -- the listing `preprocess_video_and_predict` walks, get_prob_video.py:79-90."""
    rows = np.where(records[:, 1] == track)[0]
    present = np.zeros(total_frames, dtype=bool)
    present[records[rows, 0]] = True
    frames = torch.zeros((total_frames, 224, 224, 3), dtype=torch.uint8, device=tiles.device)
    if len(rows):
        frames[torch.from_numpy(records[rows, 0]).to(tiles.device)] = tiles[torch.from_numpy(rows).to(tiles.device)]
    return frames, present
