"""CSV wire formats of the reference's per-video tables and the dataset-level fusion driver built on them.

    static__<video>.csv / dynamic__<video>.csv    7 video-order columns           get_prob_video.py:182-202
    <model_name>/<video>.csv                       7 or 8 audio-order columns + "frames" ("%06d.jpg", one row per
                                                   (window, frame) pair)          get_prob_audio_8_cl.py:103-136
    C_EXPR_DB_<modality>_sd_<weight_type>_<rule2>_<rule1>.txt   submission file   get_pred_av.py:198-334, data/utils.py:212-219

pandas does the text I/O (it IS the reference's writer, so files are byte-compatible); all arithmetic of the fusion
driver (per-frame mean, softmax, weighted sum, compound rule, argmax) runs in libavcer_hip.so.
"""
from __future__ import annotations

import os

import numpy as np
import pandas as pd

from .audio_pipeline import EMO_AUDIO_8
from .fusion import WEIGHTS_AV_1
from .video_pipeline import DICT_EMO_VIDEO

EMO_AUDIO_7 = EMO_AUDIO_8[:7]
SUBMISSION_COLUMNS = ("image_location", "Fearfully_Surprised", "Happily_Surprised", "Sadly_Surprised",
                      "Disgustedly_Surprised", "Angrily_Surprised", "Sadly_Fearful", "Sadly_Angry")  # get_pred_av.py:307-316
_RENAMES = {"135-24-1920x1080": "135-24-1920x1080_left", "6-30-1920x1080": "6-30-1920x1080_right"}  # get_prob_audio_8_cl.py:131-134


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def write_visual_csvs(static_probs, dynamic_logits, save_path: str, video_name: str):
    """get_prob_video.py:182-202.  Returns (dynamic_path, static_path)."""
    os.makedirs(save_path, exist_ok=True)
    paths = []
    for kind, table in (("dynamic", dynamic_logits), ("static", static_probs)):
        path = os.path.join(save_path, f"{kind}__{video_name}.csv")
        pd.DataFrame(_np(table), columns=list(DICT_EMO_VIDEO)).to_csv(path, index=False)
        paths.append(path)
    return tuple(paths)


def write_audio_csv(rows, frames, save_path: str, model_name: str, video_name: str) -> str:
    """get_prob_audio_8_cl.py:103-136: one row per (window, frame) pair, `frames` as "%06d.jpg"."""
    rows = _np(rows)
    cols = list(EMO_AUDIO_7 if rows.shape[1] == 7 else EMO_AUDIO_8)
    df = pd.DataFrame(rows, columns=cols)
    df["frames"] = [str(int(f)).zfill(6) + ".jpg" for f in _np(frames)]
    out_dir = os.path.join(save_path, model_name)
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"{_RENAMES.get(video_name, video_name)}.csv")
    df.to_csv(path, index=False)
    return path


def read_visual_csv(path: str) -> np.ndarray:
    return pd.read_csv(path)[list(DICT_EMO_VIDEO)].values


def read_audio_csv(path: str):
    """Returns (rows [m, C] without NaN rows, frame index [m]); `.dropna()` as get_pred_av.py:243-245."""
    df = pd.read_csv(path).dropna()
    cols = [c for c in EMO_AUDIO_8 if c in df.columns]
    frames = np.array([int(str(f).split(".")[0]) for f in df["frames"]], dtype=np.int64)
    return df[cols].values, frames


def dataset_fusion_video(engine, prediction_file_format: str, path_pred: str, name_videos, weights_1, weights_2=(1, 1),
                         modality: str = "V", weight_type: str = "single", ce_weights_type: bool = False, ce_mask: bool = True,
                         ce_mask_types=(True, False), save_path: str = "src/pred_results/DF_C_EXPR_DB/"):
    """get_pred_video.get_c_expr_db_pred (get_pred_video.py:203-342): fusion of the two visual models only.
    Writes the three submission files of the reference (fused, static-only, dynamic-only; the single-model files carry
    the script's global `ce_mask_types` list in their name, get_pred_video.py:330,338 -- reproduced as is).
    Returns (image_locations, (v_pred, s_pred, d_pred), (v_txt, s_txt, d_txt))."""
    import torch

    fmt = pd.read_csv(prediction_file_format)
    listed = set(fmt.image_location)
    stats, dyns, locations = [], [], []
    for video in name_videos:
        stat = read_visual_csv(os.path.join(path_pred, "static__" + video) + ".csv")
        dyn = read_visual_csv(os.path.join(path_pred, "dynamic__" + video) + ".csv")
        n = min(len(stat), len(dyn))  # the reference intersects the two tables' frame names with the listed ones
        names = [f"{video}/{str(f + 1).zfill(5)}.jpg" for f in range(n)]
        sel = np.array([f for f in range(n) if names[f] in listed], dtype=np.int64)
        if not len(sel):
            continue
        stats.append(torch.from_numpy(stat[sel].astype(np.float32)))
        dyns.append(torch.from_numpy(dyn[sel].astype(np.float32)))
        locations.extend(sorted(names[f] for f in sel))  # `sorted(need_image_location)`: zero-padded names keep frame order
    stat_all = torch.cat(stats).to(engine.device)
    dyn_all = torch.cat(dyns).to(engine.device)
    n = len(locations)
    zeros = torch.zeros(n, 7, device=engine.device)  # a third model with weight 0 adds exactly 0.0 to every class
    w1 = [list(weights_1[0]), list(weights_1[1]), [0.0] * 7]
    _, am_w = engine.fuse(stat_all, dyn_all, zeros, n, w1, (float(weights_2[0]), float(weights_2[1]), 0.0), ce_weights_type, ce_mask)
    _, am_raw = engine.fuse(stat_all, dyn_all, zeros, n, None, (1, 1, 1), ce_weights_type, ce_mask)  # un-weighted single models
    preds = (am_w[0].cpu().numpy(), am_raw[1].cpu().numpy(), am_raw[2].cpu().numpy())
    os.makedirs(save_path, exist_ok=True)
    tails = (f"sd_{weight_type}_{ce_weights_type}_{ce_mask}", f"static_{weight_type}_{ce_weights_type}_{list(ce_mask_types)}",
             f"dynamic_{weight_type}_{ce_weights_type}_{list(ce_mask_types)}")
    paths = tuple(os.path.join(save_path, f"C_EXPR_DB_{modality}_{t}.txt") for t in tails)
    for path, pred in zip(paths, preds):
        save_txt(SUBMISSION_COLUMNS, locations, pred, path)
    return locations, preds, paths


def save_txt(column_names, file_names, labels, save_name: str):
    """data/utils.py:212-219."""
    with open(save_name, "w") as fh:
        fh.write(",".join(column_names) + "\n")
        for f, l in zip(file_names, labels):
            fh.write(f"{f},{l}\n")


def align_video(n_video_frames: int, audio_frames: np.ndarray, image_location, video: str):
    """Index logic of get_pred_av.py:233-275 for one video.
    Returns (sel: indices of the video frames listed in the prediction file, in table order;
             aud_pos: for every selected frame the row of the per-frame audio table it uses (tail repeats the last))."""
    wanted = set(image_location)
    sel = np.array([f for f in range(n_video_frames) if f"{video}/{str(f + 1).zfill(5)}.jpg" in wanted], dtype=np.int64)
    have = np.unique(audio_frames)
    aud_sel = np.array([f for f in have if f"{video}/{str(int(f) + 1).zfill(5)}.jpg" in wanted], dtype=np.int64)
    if len(sel) and not len(aud_sel):
        raise IndexError("index -1 is out of bounds for axis 0 with size 0")  # curr_pred_audio[-1], get_pred_av.py:266
    aud_pos = np.minimum(np.arange(len(sel)), len(aud_sel) - 1)
    return sel, aud_sel, aud_pos


def dataset_fusion(engine, prediction_file_format: str, root: str, path_preds, name_videos, weights_1=WEIGHTS_AV_1,
                   weights_2=(1, 1, 1), modality: str = "av", weight_type: str = "w", ce_weights_type: bool = False,
                   ce_mask: bool = True, save_path: str = "src/pred_results/DF_C_EXPR_DB/"):
    """get_pred_av.get_c_expr_db_pred (get_pred_av.py:198-334): read the per-video CSVs, keep the frames the challenge's
    prediction file lists, fuse, write the submission txt.  Returns (image_locations, av_pred, txt_path)."""
    import torch

    fmt = pd.read_csv(prediction_file_format)
    by_video = {}
    for loc in fmt.image_location:
        by_video.setdefault(loc.split("/")[0], []).append(loc)
    stats, dyns, auds, locations = [], [], [], []
    for video in name_videos:
        stat = read_visual_csv(os.path.join(root, path_preds[0], "static__" + video) + ".csv")
        dyn = read_visual_csv(os.path.join(root, path_preds[0], "dynamic__" + video) + ".csv")
        rows, frames = read_audio_csv(os.path.join(root, path_preds[1], path_preds[2], video) + ".csv")
        image_location = by_video.get(video, [])
        sel, aud_sel, aud_pos = align_video(len(stat), frames, image_location, video)
        if not len(sel):
            continue
        # per-frame mean of the window rows on the GPU: every CSV row covers exactly one frame
        n_cov = int(frames.max()) + 1
        mean, _ = engine.audio_frame_mean(rows.astype(np.float32), frames, frames + 1, n_cov)
        aud_rows = mean.index_select(0, torch.from_numpy(aud_sel[aud_pos]).to(mean.device))
        if aud_rows.shape[1] < 8:  # 7-class model: the fuse kernel only reads the first 7 columns
            aud_rows = aud_rows.contiguous()
        stats.append(torch.from_numpy(stat[sel].astype(np.float32)))
        dyns.append(torch.from_numpy(dyn[sel].astype(np.float32)))
        auds.append(aud_rows)
        locations.extend(f"{video}/{str(f + 1).zfill(5)}.jpg" for f in sel)
    stat_all = torch.cat(stats).to(engine.device)
    dyn_all = torch.cat(dyns).to(engine.device)
    aud_all = torch.cat(auds).contiguous()
    _, am = engine.fuse(stat_all, dyn_all, aud_all, len(locations), weights_1, weights_2, ce_weights_type, ce_mask)
    av_pred = am[0].cpu().numpy()
    os.makedirs(save_path, exist_ok=True)
    txt = os.path.join(save_path, f"C_EXPR_DB_{modality}_sd_{weight_type}_{ce_weights_type}_{ce_mask}.txt")
    save_txt(SUBMISSION_COLUMNS, locations, av_pred, txt)
    return locations, av_pred, txt
