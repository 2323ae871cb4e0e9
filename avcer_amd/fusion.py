"""Array counterpart of run.get_c_expr_db_pred (run.py:25-189): no pandas, no files; constants of run.py:316-344."""
from __future__ import annotations

import numpy as np
import torch

from .engine import Engine

# run.py:316-344 (== get_weights_matrices.py:51-59 transposed); rows VS, VD, A; columns in audio order
WEIGHTS_AV_1 = (
    (0.89900098, 0.10362151, 0.08577635, 0.04428126, 0.89679865, 0.02656456, 0.63040305),
    (0.01223291, 0.21364307, 0.66688002, 0.93791526, 0.0398964, 0.48670648, 0.22089692),
    (0.08876611, 0.68273542, 0.24734363, 0.01780348, 0.06330495, 0.48672896, 0.14870002),
)
# The other two learned tables of get_weights_matrices.py (rows 0-6 transposed: one row per model, columns in audio
# order; row 7 of each matrix = the level-2 "double" model weights that get_pred_*.py pass as weights_2):
#   get_weights_matrices.py:5-16   video-only fusion (VS, VD), used by get_pred_video.get_c_expr_db_pred
#   get_weights_matrices.py:28-39  7-class audio + video ("Acl7": ExprModelV2, padding "repeat", step 1; get_pred_av.py:362-365)
#   get_weights_matrices.py:51-62  8-class audio + video ("Acl8"), level 1 == WEIGHTS_AV_1 above
WEIGHTS_V_1 = (
    (0.42633145, 0.57803352, 0.01878466, 0.86451425, 0.16464752, 0.03786653, 0.81048546),
    (0.57366855, 0.42196648, 0.98121534, 0.13548575, 0.83535248, 0.96213347, 0.18951454),
)
WEIGHTS_V_2 = (0.36499999999999994, 0.22999999999999998)
WEIGHTS_AV7_1 = (
    (0.85806901, 0.2579578, 0.2579578, 0.72010502, 0.62082661, 0.06281922, 0.70875895),
    (0.11491265, 0.46222294, 0.62411413, 0.16716238, 0.31962795, 0.16603196, 0.24433032),
    (0.02701833, 0.27981925, 0.17148297, 0.1127326, 0.05954545, 0.77114883, 0.04691073),
)
WEIGHTS_AV7_2 = (0.060000000000000005, 0.21000000000000002, 0.01)
WEIGHTS_AV_2 = (0.16000000000000003, 0.36000000000000004, 0.01)
COMPOUND_NAMES = ("Fearfully Surprised", "Happily Surprised", "Sadly Surprised", "Disgustedly Surprised",
                  "Angrily Surprised", "Sadly Fearful", "Sadly Angry")  # run.py:66-74
MODEL_ORDER = ("AV", "VS", "VD", "A")


def covered_frames(frame_lo, frame_hi, n_frames: int) -> int:
    """Number of leading video frames that have at least one audio window (run.py:90-103: frames present in the
    group-by, restricted to frames of the video; the rest repeat the last audio row)."""
    lo, hi = np.asarray(frame_lo), np.asarray(frame_hi)
    cover = np.zeros(n_frames, dtype=bool)
    for a, b in zip(lo, hi):
        cover[max(int(a), 0):max(min(int(b), n_frames), 0)] = True
    n_aud = int(cover.sum())
    if n_aud == 0:
        raise IndexError("index -1 is out of bounds for axis 0 with size 0")  # `audio_df[-1]`, run.py:100
    if not cover[:n_aud].all():
        raise ValueError("audio windows do not cover a contiguous prefix of the video frames")
    return n_aud


def fuse(engine: Engine, stat_probs, dyn_logits, win_logits, frame_lo, frame_hi, weights_1=WEIGHTS_AV_1,
         weights_2=(1, 1, 1), ce_weights_type: bool = False, ce_mask: bool = True):
    """One clip/video.  stat_probs, dyn_logits [n,7] in video column order; win_logits [n_win, C] raw audio logits
    with their frame spans.  Returns (comp_prob f64 [4,n,7], comp_argmax i32 [4,n]) ordered AV, VS, VD, A."""
    n = int(stat_probs.shape[0])
    n_aud = covered_frames(frame_lo, frame_hi, n)
    mean, _ = engine.audio_frame_mean(win_logits, frame_lo, frame_hi, n)
    return engine.fuse(stat_probs, dyn_logits, mean, n_aud, weights_1, weights_2, ce_weights_type, ce_mask)


def fuse_clips(engine: Engine, stat_probs, dyn_logits, clip_audio_logits, weights_1=WEIGHTS_AV_1,
               weights_2=(1, 1, 1), ce_weights_type: bool = False, ce_mask: bool = True):
    """Batch of N clips of T frames with ONE audio window each covering all T frames (the benchmark clip of
    BASELINE.json: 16 frames + 2 s of audio).  stat/dyn [N,T,7], clip_audio_logits [N,C].
    The per-frame mean of a single window is the window itself, so the rows are just repeated T times."""
    n, t = int(stat_probs.shape[0]), int(stat_probs.shape[1])
    aud = clip_audio_logits[:, None, :].expand(n, t, clip_audio_logits.shape[-1]).reshape(n * t, -1)
    prob, am = engine.fuse(stat_probs.reshape(n * t, 7), dyn_logits.reshape(n * t, 7), aud, n * t, weights_1,
                           weights_2, ce_weights_type, ce_mask)
    return prob.view(4, n, t, 7), am.view(4, n, t)
