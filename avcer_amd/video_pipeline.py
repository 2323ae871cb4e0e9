"""Batched counterpart of get_prob_video.preprocess_video_and_predict (get_prob_video.py:67-204).

The reference walks frames one by one; its rules only decide WHICH feature rows form each LSTM window and WHICH
result row each frame reports.  `plan_clip` restates those rules as index arithmetic on the host (no tensor data),
the GPU then runs the static CNN once over all present frames, gathers the windows, runs the LSTM once, and the
per-frame tables are assembled by row gathers.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np
import torch

from .engine import Engine, MODE_DEFAULT

DICT_EMO_VIDEO = ("Neutral", "Happiness", "Sadness", "Surprise", "Fear", "Disgust", "Anger")  # get_prob_video.py:56-64


def lstm_step(fps: float) -> int:
    """get_prob_video.py:77 (Python's round: banker's rounding)."""
    return round((5 * fps) / 25)


@dataclass
class ClipPlan:
    static_src: list = field(default_factory=list)  # per frame: row of the feature/prob table, -1 = zeros
    dyn_src: list = field(default_factory=list)     # per frame: row of the LSTM output table, -1 = zeros
    windows: list = field(default_factory=list)     # per LSTM evaluation: 10 feature-table rows


def plan_clip(present, fps: float, feat_base: int = 0, win_base: int = 0) -> ClipPlan:
    """Hold / zero / reset rules of get_prob_video.py:91-178 for one clip.  `present[i]` = a face crop exists."""
    step = lstm_step(fps)
    if step <= 0:
        raise ZeroDivisionError("integer division or modulo by zero")  # `curr_idx_frame % step`, get_prob_video.py:114
    plan = ClipPlan()
    window: list[int] = []
    last = None
    nfeat = 0
    for i, p in enumerate(present):
        if p:
            s = feat_base + nfeat
            nfeat += 1
            if i % step == 0:
                window = [s] * 10 if not window else window[1:] + [s]   # :117-120
                plan.windows.append(list(window))
                last = win_base + len(plan.windows) - 1
                d = last
            else:
                d = last if last is not None else -1                      # :157-162
            plan.static_src.append(s)
            plan.dyn_src.append(d)
        else:
            window = []                                                   # :169
            if last is not None:                                          # :170-173
                plan.static_src.append(plan.static_src[-1])
                plan.dyn_src.append(plan.dyn_src[-1])
            else:                                                         # :175-178
                plan.static_src.append(-1)
                plan.dyn_src.append(-1)
    return plan


class _DevicePlan:
    """Index tensors of one (present mask, fps) combination, resident on the device.  Building them costs Python loops
    and host->device copies that serialise behind in-flight GPU work, so they are cached per mask."""

    def __init__(self, engine: Engine, present: np.ndarray, fps: float):
        n, t = present.shape
        plans, fb, wb = [], 0, 0
        for c in range(n):
            p = plan_clip(present[c], fps, fb, wb)
            fb += int(present[c].sum())
            wb += len(p.windows)
            plans.append(p)
        dev = engine.device
        self.n_feat, self.n_win = fb, wb
        self.sel = None if fb == n * t else torch.from_numpy(np.nonzero(present.reshape(-1))[0]).to(dev)
        s_src = np.array([i if i >= 0 else fb for p in plans for i in p.static_src], dtype=np.int64)
        d_src = np.array([i if i >= 0 else wb for p in plans for i in p.dyn_src], dtype=np.int64)
        win = np.array([w for p in plans for w in p.windows], dtype=np.int32).reshape(-1, 10)
        assert win.size == 0 or (win.min() >= 0 and win.max() < max(fb, 1))
        self.s_src = torch.from_numpy(s_src).to(dev)
        self.d_src = torch.from_numpy(d_src).to(dev)
        self.win = torch.from_numpy(win).to(dev)
        self.zero_row = torch.zeros(1, 7, device=dev)


def _device_plan(engine: Engine, present: np.ndarray, fps: float) -> _DevicePlan:
    """Index plans live ON the Engine object (they hold tensors of its device), so they die with it."""
    cache = engine.__dict__.setdefault("_plan_cache", {})
    key = (present.shape, present.tobytes(), float(fps))
    plan = cache.get(key)
    if plan is None:
        if len(cache) > 64:
            cache.clear()
        plan = cache[key] = _DevicePlan(engine, present, fps)
    return plan


def visual_forward(engine: Engine, frames_u8: torch.Tensor, present, fps: float, mode: int = MODE_DEFAULT):
    """frames_u8 [N,T,H,W,3] (or [T,H,W,3]) RGB tiles, present [N,T] bool.
    Returns (static_probs [N,T,7], dynamic_logits [N,T,7]) float32 in VIDEO column order (DICT_EMO_VIDEO).
    The reference's tables turn float64 when they contain a zero placeholder row; the values are the same."""
    single = frames_u8.dim() == 4
    if single:
        frames_u8 = frames_u8[None]
    present = np.ascontiguousarray(np.asarray(present, dtype=bool).reshape(frames_u8.shape[0], frames_u8.shape[1]))
    n, t = present.shape
    plan = _device_plan(engine, present, fps)
    dev = engine.device
    flat = frames_u8.reshape(n * t, *frames_u8.shape[2:])
    if plan.n_feat:
        frames_sel = flat.to(dev) if plan.sel is None else flat.to(dev).index_select(0, plan.sel)
        _, probs, feats = engine.static_forward(frames_sel, mode)
        stat = torch.cat([probs, plan.zero_row]).index_select(0, plan.s_src)
        if plan.n_win:
            dl = engine.dynamic_forward(engine.gather_windows(feats, plan.win, validated=True), mode)
            dyn = torch.cat([dl, plan.zero_row]).index_select(0, plan.d_src)
        else:
            dyn = torch.zeros(n * t, 7, device=dev)
    else:
        stat = torch.zeros(n * t, 7, device=dev)
        dyn = torch.zeros(n * t, 7, device=dev)
    stat, dyn = stat.view(n, t, 7), dyn.view(n, t, 7)
    return (stat[0], dyn[0]) if single else (stat, dyn)


def read_face_dir(path_images: str, total_frames: int, track: str = "00"):
    """The file side of `preprocess_video_and_predict` (get_prob_video.py:79-100): the face crops stage 0 wrote as
    `<path_images>/<track>/NNNNNN.jpg`, one per frame in which the track was seen.  Returns (frames u8 [total_frames,224,224,3] RGB,
    present bool [total_frames]): frame i is the JPEG `%06d.jpg` decoded to RGB and resized to 224 x 224 with PIL's NEAREST filter --
    the very call `pth_processing` makes (data/utils.py:34) -- or zeros where the file does not exist.  PIL decodes here where the
    reference decodes with cv2.imread; both sit on libjpeg, a decoder difference of a grey level cannot be excluded.
    `os.listdir` of a missing track directory raises FileNotFoundError as in the reference."""
    from PIL import Image

    folder = os.path.join(path_images, track)
    names = set(os.listdir(folder))
    frames = np.zeros((total_frames, 224, 224, 3), dtype=np.uint8)
    present = np.zeros(total_frames, dtype=bool)
    for i in range(total_frames):
        name = str(i).zfill(6) + ".jpg"
        if name in names:
            with Image.open(os.path.join(folder, name)) as img:
                frames[i] = np.asarray(img.convert("RGB").resize((224, 224), Image.Resampling.NEAREST))
            present[i] = True
    return frames, present


def preprocess_video_and_predict(engine: Engine, path_images: str = "", save_path: str = "", fps: float = 30, total_frames: int = 0,
                                 flag_save_prob: bool = False, mode: int = MODE_DEFAULT):
    """`get_prob_video.preprocess_video_and_predict` (get_prob_video.py:67-204) with the reference's argument meaning, on the HIP
    path: the face-crop directory of one video in, the two per-frame tables out -- (dynamic logits, static probabilities), float32
    [total_frames, 7] in DICT_EMO_VIDEO column order -- and `dynamic__<video>.csv` / `static__<video>.csv` under `save_path` when
    `flag_save_prob` (the reference's files, io_formats.write_visual_csvs).  Heat maps (`flag_heatmaps`) need a backward pass and
    are not part of this build."""
    from . import io_formats

    frames, present = read_face_dir(path_images, total_frames)
    stat, dyn = visual_forward(engine, torch.from_numpy(frames), present, fps, mode)
    if flag_save_prob:
        io_formats.write_visual_csvs(stat, dyn, save_path, os.path.basename(path_images))
    return dyn.cpu().numpy(), stat.cpu().numpy()
