"""`run.run_inference` (src/run.py:190-303) on in-memory inputs: face detection and tracking, visual models on the
first track, audio model over sliding windows, compound-expression fusion.

What the reference does through files -- cv2.VideoCapture frames, JPEG crops under `<save>/<video>/00/`, an ffmpeg
wav at 16 kHz, CSV tables when `flag_save_prob` -- is replaced by arrays: decoded BGR frames `[T,H,W,3]` u8 and a mono
waveform at 16 kHz go in; per-frame predictions come out.  Plotting and Grad-CAM heat maps are not part of this build.
"""
from __future__ import annotations

import os
import time
from typing import Optional, Sequence

import numpy as np
import torch

from . import io_formats
from .audio_pipeline import audio_forward, replicate_per_frame
from .engine import MODE_DEFAULT
from .face_tiles import VideoTiler, track_clip
from .fusion import MODEL_ORDER, fuse
from .video_pipeline import visual_forward


def run_inference(engine, frames_bgr, wav, fps: float, detector=None, detections: Optional[Sequence[np.ndarray]] = None,
                  path_save_results: str = "", name_video: str = "video", flag_save_prob: bool = False,
                  weights_prob_model=None, weights_model=(1, 1, 1), ce_weights_type: bool = True, ce_mask: bool = False,
                  sr: int = 16000, window: float = 4, step: float = 0.5, padding: str = "mean", mode: int = MODE_DEFAULT):
    """engine: an `Engine` with the static, dynamic and audio weights loaded.  frames_bgr u8 [T,H,W,3] as cv2 decodes
    them; wav float32 [L] mono at `sr`; fps as `int(cv2.CAP_PROP_FPS)` gives it (get_face_images.py:23).
    `detector`: a `face_tiles.RetinaFacePredictor` (threshold 0.8 in the reference); or pass per-frame `detections`.
    Defaults follow `run_inference`'s signature (Rule 2 weights on, Rule 1 mask off; `run.py --help` flips them).
    Returns a dict: av / vs / vd / a predictions (int32 [T], compound class per frame), `compound_prob` f64 [4,T,7],
    `static_probs`, `dynamic_logits` [T,7], `audio_rows` / `audio_frames` (the audio table), `records` (face files),
    `real_time_factor` (elapsed / video duration, the figure run.py:307 prints)."""
    start_time = time.time()                                                # run.py:200
    frames = frames_bgr if torch.is_tensor(frames_bgr) else torch.from_numpy(np.ascontiguousarray(frames_bgr))
    total_frames = int(frames.shape[0])
    # the duration the real-time factor divides by, settled before any work is queued: `int(cv2.CAP_PROP_FPS)` is 0 on a
    # broken container, and a finished prediction must not be lost to a ZeroDivisionError in the last statement
    duration = total_frames / fps if (fps and fps > 0 and total_frames > 0) else None
    if detections is None and detector is None:
        raise ValueError("give a detector or the per-frame detections")
    wav_t = wav if torch.is_tensor(wav) else torch.from_numpy(np.ascontiguousarray(wav, dtype=np.float32))
    dev = engine.device
    wav_t = wav_t.to(dev)
    main = torch.cuda.current_stream(dev)
    side = engine.__dict__.get("_side_stream")
    if side is None:
        side = engine.__dict__["_side_stream"] = torch.cuda.Stream(dev)
    host = {}  # detector / tracker / crop results: independent of the arithmetic mode, computed once

    def gpu_work(m):
        # (1) the audio branch depends on nothing of the visual one: it is queued FIRST, on its own stream, and runs while
        #     the host walks the tracker loop below (750 numpy + linear_sum_assignment iterations for a 30 s video)
        side.wait_stream(main)
        wav_t.record_stream(side)  # allocated on `main`: the allocator must not hand it out while `side` still reads it
        joined = False
        try:
            with torch.cuda.stream(side):
                win_logits, lo, hi = audio_forward(engine, wav_t, sr, fps, window, step, padding, m)  # get_prob_audio_8_cl.py:68-138
            # (2) faces -> tracks -> tiles (get_face_images.py:38-63), then the visual models on track 00
            if "clip" not in host:
                dets = detections if detections is not None else detector.batch(frames, rgb=False)  # get_face_images.py:49
                records, tiles = VideoTiler(engine).process(frames, dets)
                if not (len(records) and (records[:, 1] == 0).any()):
                    raise FileNotFoundError("no face track 00 (os.listdir(<faces>/00) fails in the reference, get_prob_video.py:79)")
                host["records"] = records
                host["clip"] = track_clip(records, tiles, 0, total_frames)
            clip, present = host["clip"]
            static_probs, dynamic_logits = visual_forward(engine, clip, present, fps, m)             # get_prob_video.py:67-204
            # (3) fusion last, behind both branches; nothing has been copied to the host yet
            main.wait_stream(side)
            joined = True
            win_logits.record_stream(main)
            prob, am = fuse(engine, static_probs, dynamic_logits, win_logits, lo, hi, weights_prob_model, weights_model,
                            ce_weights_type, ce_mask)                                                # run.py:25-189
            return static_probs, dynamic_logits, win_logits, lo, hi, prob, am
        finally:
            # the reference's failure paths (no face track, a detector error) unwind from here: the audio branch already queued
            # on `side` is joined all the same, so that no launch of this video outlives the call (its workspace and the
            # range-contract counter belong to the next one)
            if not joined:
                main.wait_stream(side)

    # MODE_F16X3: one read of the range-contract counter behind the last launch; a video during which an activation left fp16's
    # range is run again in MODE_FP32 (engine.guarded)
    static_probs, dynamic_logits, win_logits, lo, hi, prob, am = engine.guarded(mode, gpu_work)
    records = host["records"]
    rows, aud_frames = replicate_per_frame(win_logits.cpu().numpy(), lo, hi)
    if flag_save_prob:
        io_formats.write_visual_csvs(static_probs, dynamic_logits, path_save_results, name_video)
        io_formats.write_audio_csv(rows, aud_frames, path_save_results, "audio", name_video)
    am = am.cpu().numpy()
    out = {name.lower(): am[i] for i, name in enumerate(MODEL_ORDER)}
    out.update(compound_prob=prob.cpu().numpy(), static_probs=static_probs.cpu().numpy(),
               dynamic_logits=dynamic_logits.cpu().numpy(), audio_rows=rows, audio_frames=aud_frames, records=records)
    # "Real-time factor for compound expression prediction" as run.py:304-307 prints it: elapsed / video duration (the
    # device -> host copies above have synchronised the stream, so the clock covers all the work); None where the
    # container reported no frame rate
    out["real_time_factor"] = (time.time() - start_time) / duration if duration else None
    return out
