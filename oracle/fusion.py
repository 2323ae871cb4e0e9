"""ORACLE (test infrastructure, never shipped on the product path).

Numpy restatement of AVCER's probability fusion (src/run.py:25-189, src/data/utils.py:125-127,222-241),
array-only (no pandas, no files).  Pinned against the imported reference's `run.get_c_expr_db_pred`
and `data.utils.get_compound_expression` via tests/golden/fusion.npz.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np

# audio-model column order is the fusion order (run.py:56-65); video tables use get_prob_video.py:56-64
VIDEO_TO_AUDIO_ORDER = (0, 6, 5, 4, 1, 2, 3)
# run.py:66-74, indices in audio order
COMPOUND_PAIRS = ((3, 6), (4, 6), (5, 6), (2, 6), (1, 6), (3, 5), (1, 5))
DICT_WEIGHTS = {1: 5, 2: 6, 3: 5, 4: 6, 5: 4, 6: 2}  # run.py:116-123
# run.py:316-344 (== get_weights_matrices.py:51-59 transposed); rows VS, VD, A
WEIGHTS_AV_1 = (
    (0.89900098, 0.10362151, 0.08577635, 0.04428126, 0.89679865, 0.02656456, 0.63040305),
    (0.01223291, 0.21364307, 0.66688002, 0.93791526, 0.0398964, 0.48670648, 0.22089692),
    (0.08876611, 0.68273542, 0.24734363, 0.01780348, 0.06330495, 0.48672896, 0.14870002),
)


def softmax(matrix):
    """data/utils.py:125-127."""
    exp_matrix = np.exp(matrix - np.max(matrix, axis=1, keepdims=True))
    return exp_matrix / np.sum(exp_matrix, axis=1, keepdims=True)


def get_compound_expression(pred, ce_weights_type: bool, ce_mask: bool):
    """data/utils.py:222-241 with com_emo / dict_weights of run.py:66-74,116-123."""
    pred = np.asarray(pred)
    prob = np.zeros((len(pred), len(COMPOUND_PAIRS)))
    for idx, (i1, i2) in enumerate(COMPOUND_PAIRS):
        if ce_weights_type:
            s_w = DICT_WEIGHTS[i1] + DICT_WEIGHTS[i2]
            w1, w2 = DICT_WEIGHTS[i1] / s_w, DICT_WEIGHTS[i2] / s_w
        else:
            w1, w2 = 1, 1
        if ce_mask:
            pred = np.where(pred > 1 / 7, pred, 0)
        prob[:, idx] = pred[:, i1] * w1 + pred[:, i2] * w2
    return prob


def audio_per_frame(aud_rows: np.ndarray, aud_frames: np.ndarray, n_frames: int) -> np.ndarray:
    """run.py:90-103: group-by-frame MEAN OF LOGITS (pandas keeps float32), keep frames that exist in the
    video (frame f -> image f+1 <= n_frames), drop 'Other', softmax over the remaining 7, pad the tail with
    the last row.  Returns [n_frames, 7] (fewer rows never happen unless there is no audio row at all)."""
    frames = np.asarray(aud_frames)
    uniq = np.unique(frames)
    uniq = uniq[(uniq >= 0) & (uniq < n_frames)]
    rows = []
    for f in uniq:
        sel = aud_rows[frames == f]
        rows.append(sel.astype(np.float64).mean(axis=0).astype(aud_rows.dtype))
    aud = softmax(np.stack(rows)[:, :7])
    if n_frames > len(aud):
        aud = np.vstack((aud, [aud[-1]] * (n_frames - len(aud))))
    return aud


def fuse(stat_probs, dyn_logits, aud_rows, aud_frames, weights_1=WEIGHTS_AV_1, weights_2=(1, 1, 1),
         ce_weights_type: bool = False, ce_mask: bool = True):
    """run.get_c_expr_db_pred (run.py:25-165) on arrays.

    stat_probs [n,7] (already softmaxed) and dyn_logits [n,7] in VIDEO column order; aud_rows [m,8] raw logits
    with their frame indices.  Returns (comp_prob [4,n,7] float64 in the order AV, VS, VD, A; comp_argmax [4,n])."""
    order = list(VIDEO_TO_AUDIO_ORDER)
    n = len(dyn_logits)
    stat = np.asarray(stat_probs)[:, order]
    dyn = softmax(np.asarray(dyn_logits)[:, order])
    aud = audio_per_frame(np.asarray(aud_rows), aud_frames, n)
    preds = [stat, dyn, aud]
    if weights_1:
        weighted = [preds[i] * list(weights_1[i]) * weights_2[i] for i in range(3)]
        final = weighted[0] + weighted[1] + weighted[2]
        parts = weighted
    else:
        final = np.sum(preds, axis=0) / 3
        parts = preds
    probs = [get_compound_expression(p, ce_weights_type, ce_mask) for p in (final, *parts)]
    comp = np.stack(probs)
    return comp, np.argmax(comp, axis=2)
