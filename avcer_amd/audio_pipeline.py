"""Batched counterpart of EmotionRecognition.load_audio_features (get_prob_audio_8_cl.py:68-138).

Index arithmetic (window starts, frame spans, Python banker's rounding) is restated on the host; slicing, padding,
normalisation and the model run on the GPU in one batch over all windows of a waveform.
"""
from __future__ import annotations

import numpy as np
import torch

from .engine import Engine, MODE_DEFAULT

EMO_AUDIO_8 = ("Neutral", "Anger", "Disgust", "Fear", "Happiness", "Sadness", "Surprise", "Other")  # :114-123


def chunk_spans(n_samples: int, sr: int, fps: float, window: float, step: float):
    """get_prob_audio_8_cl.py:70-99.  Returns int arrays (start, end, frame_lo, frame_hi), one row per window;
    window i reports its logits for frames range(frame_lo[i], frame_hi[i])."""
    window_a = int(window * sr)
    step_a = int(step * sr)
    rows = []
    for start in range(0, n_samples + 1, step_a):
        end = min(start + window_a, n_samples)
        rows.append((start, end, round(start / sr * fps), round(end / sr * fps + 1)))
    a = np.asarray(rows, dtype=np.int64).reshape(-1, 4)
    return a[:, 0], a[:, 1], a[:, 2], a[:, 3]


def audio_forward(engine: Engine, wav: torch.Tensor, sr: int = 16000, fps: float = 25, window: float = 4,
                  step: float = 0.5, padding: str = "mean", mode: int = MODE_DEFAULT):
    """wav f32 [L] (mono, already at `sr`).  Returns (window_logits [n_win, C], frame_lo [n_win], frame_hi [n_win]).
    An empty tail window (len(wav) % (step*sr) == 0) yields NaN logits, as in the reference ('mean' padding of an
    empty chunk is NaN, data/utils.py:76-82)."""
    if padding not in ("mean", "constant", "repeat"):
        raise ValueError(f"padding={padding!r}")
    wav = wav.reshape(-1)
    starts, ends, lo, hi = chunk_spans(int(wav.numel()), sr, fps, window, step)
    chunks = engine.audio_chunks(wav, starts, ends, int(window * sr), padding)
    logits = engine.audio_forward(chunks, normalize=True, mode=mode)
    return logits, lo, hi


def replicate_per_frame(logits: np.ndarray, lo, hi):
    """get_prob_audio_8_cl.py:94-101: the reference's DataFrame content (rows, frame index per row)."""
    rows, frames = [], []
    for lg, a, b in zip(logits, lo, hi):
        for f in range(int(a), int(b)):
            rows.append(lg)
            frames.append(f)
    if not rows:
        return np.zeros((0, logits.shape[1]), logits.dtype), np.zeros((0,), np.int64)
    return np.stack(rows), np.asarray(frames, dtype=np.int64)
