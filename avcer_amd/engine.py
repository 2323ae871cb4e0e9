"""Thin object wrapper over the C ABI: one Engine = one avcer_ctx on one GPU.

torch is used for device memory (tensor.data_ptr()) and the current HIP stream only; every arithmetic step of
the hot path runs inside libavcer_hip.so.  All methods raise AvcerError on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, packing
from ._lib import SPLIT_TRAILER, AvcerError, ConvDesc

MODE_FP32 = 0    # exact f32 FMA chains on the f32 MFMA: the reference's own arithmetic, a third of the speed
MODE_BF16 = 1    # plain bf16 operands: throughput only, misses the 1e-4 parity gate
MODE_F16X3 = 2   # fp16 hi/lo operand pairs, 3 MFMAs per product: f32-grade results (include/avcer_hip.h)
# What every host mirror runs unless told otherwise.  Both parity-grade modes sit at the same distance from the reference's
# CPU path (worst |dprob| 1.6e-5 / 1.5e-5 at 8 x the synthetic logit scale, tests/test_gpu_parity_breadth.py); the fast one
# is the default, MODE_FP32 stays for checkpoints whose activations leave fp16's range (|x| >= 65504 -> NaN outputs).
MODE_DEFAULT = MODE_F16X3
PAD_MODES = {"mean": 0, "constant": 1, "repeat": 2}


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("avcer_amd needs a ROCm GPU (gfx950); there is no CPU path")
        self.lib = _lib.load()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        torch.zeros(1, device=self.device)  # make sure the primary context exists before the library uses it
        ctx = _lib.c_ctx()
        rc = self.lib.avcer_ctx_create(device, C.byref(ctx))
        if rc != 0:
            raise AvcerError(rc, "avcer_ctx_create failed")
        self.ctx = ctx
        self.audio_classes = 0
        self.x3_fallbacks = 0  # calls `guarded` had to repeat in MODE_FP32 (the x3 range contract was broken)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.avcer_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _check(self, rc: int):
        if rc != 0:
            raise AvcerError(rc, self.lib.avcer_last_error(self.ctx).decode("utf-8", "replace"))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t, dtype):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        return t.to(device=self.device, dtype=dtype).contiguous()

    def _new(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------ x3 range contract
    def x3_overflow_count(self, reset: bool = True) -> int:
        """How many GPU threads have turned a finite activation of magnitude >= 65520 into an infinite fp16 half since the
        last reset (include/avcer_hip.h avcer_x3_overflow_count).  Waits for the current stream."""
        n = C.c_int64(0)
        self._check(self.lib.avcer_x3_overflow_count(self.ctx, int(reset), C.byref(n), self._stream()))
        return int(n.value)

    def x3_overflow_clear(self):
        """Asynchronous reset of the range-contract counter, in stream order on the current stream (no wait)."""
        self._check(self.lib.avcer_x3_overflow_count(self.ctx, 1, None, self._stream()))

    def guarded(self, mode: int, call):
        """`call(mode)` with the x3 mode's range contract enforced.  `guarded` OWNS the counter for the duration of the call: in
        MODE_F16X3 it is cleared in stream order in front of the call (counts left behind by earlier unguarded kernel-level
        calls, by a call that raised or by abandoned side-stream work are not charged to this one), read ONCE behind it -- the
        read waits for the current stream, so `call` must have joined any side stream it used -- and, when an activation left
        fp16's range, the same call is repeated in MODE_FP32 (no range limit; a fresh call on the same context).  What comes
        back is then what the reference's fp32 path computes -- NaN only where the input was NaN (the empty audio window).
        Other modes pass through.  Cost: one host synchronisation per guarded call."""
        if mode != MODE_F16X3:
            return call(mode)
        self.x3_overflow_clear()
        out = call(mode)
        if self.x3_overflow_count(reset=True):
            self.x3_fallbacks += 1
            out = call(MODE_FP32)
        return out

    # ------------------------------------------------------------------ weights
    def _load(self, fn, tensors):
        blob = packing.to_blob(tensors)
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        self._check(fn(self.ctx, C.cast(buf, C.c_void_p), len(blob)))

    def load_static(self, state_dict):
        self._load(self.lib.avcer_load_static, packing.pack_static(state_dict))

    def load_dynamic(self, state_dict):
        self._load(self.lib.avcer_load_dynamic, packing.pack_dynamic(state_dict))

    def load_audio(self, state_dict):
        self._load(self.lib.avcer_load_audio, packing.pack_audio(state_dict))
        self.audio_classes = self.lib.avcer_audio_num_classes(self.ctx)

    def load_face(self, state_dict):
        self._load(self.lib.avcer_load_face, packing.pack_face(state_dict))

    def face_forward(self, frames_u8, mode: int = MODE_DEFAULT, rgb: bool = False):
        """frames u8 [N,H,W,3] (BGR unless rgb) -> (loc [N,P,4], conf [N,P,2] softmaxed, landms [N,P,10])."""
        x = self._dev(frames_u8, torch.uint8)
        if x.dim() != 4 or x.shape[-1] != 3:
            raise ValueError(f"frames must be [N,H,W,3] uint8, got {tuple(x.shape)}")
        n, h, w = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
        p = int(self.lib.avcer_face_num_priors(h, w))
        loc, conf, lm = self._new(n, p, 4), self._new(n, p, 2), self._new(n, p, 10)
        self._check(self.lib.avcer_face_forward(self.ctx, _ptr(x), n, h, w, 1 if rgb else 0, mode, _ptr(loc), _ptr(conf),
                                                _ptr(lm), self._stream()))
        return loc, conf, lm

    def set_static_batch(self, frames: int, back: int | None = None):
        """Frames per front pass of the static CNN (1..1024) and, optionally, per back pass (1..2048; 0 = two front passes)."""
        self._check(self.lib.avcer_set_static_batch(self.ctx, int(frames)))
        if back is not None:
            self._check(self.lib.avcer_set_static_back_batch(self.ctx, int(back)))

    def set_static_lanes(self, lanes: int, min_frames: int | None = None, max_frames: int | None = None):
        """2 (default): mid-sized static-CNN calls run as two half-batches on two streams; 1: always serial.  min / max frames:
        the call sizes that take the two-lane schedule."""
        self._check(self.lib.avcer_set_static_lanes(self.ctx, int(lanes)))
        if min_frames is not None or max_frames is not None:
            self._check(self.lib.avcer_set_static_lane_range(self.ctx, int(min_frames or 128), int(max_frames or 512)))

    # ------------------------------------------------------------------ forward passes
    def static_forward(self, frames_u8, mode: int = MODE_DEFAULT):
        """frames u8 [N,H,W,3] RGB -> (logits [N,7], probs [N,7], feats [N,512] pre-ReLU)."""
        x = self._dev(frames_u8, torch.uint8)
        if x.dim() != 4 or x.shape[-1] != 3:
            raise ValueError(f"frames must be [N,H,W,3] uint8, got {tuple(x.shape)}")
        n, h, w = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
        logits, probs, feats = self._new(n, 7), self._new(n, 7), self._new(n, 512)
        self._check(self.lib.avcer_static_forward(self.ctx, _ptr(x), n, h, w, mode, _ptr(logits), _ptr(probs),
                                                  _ptr(feats), self._stream()))
        return logits, probs, feats

    def static_forward_nchw(self, x, mode: int = MODE_DEFAULT):
        x = self._dev(x, torch.float32)
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 224, 224):
            raise ValueError(f"input must be [N,3,224,224] float32, got {tuple(x.shape)}")
        n = int(x.shape[0])
        logits, probs, feats = self._new(n, 7), self._new(n, 7), self._new(n, 512)
        self._check(self.lib.avcer_static_forward_nchw(self.ctx, _ptr(x), n, mode, _ptr(logits), _ptr(probs),
                                                       _ptr(feats), self._stream()))
        return logits, probs, feats

    def gather_windows(self, feats, idx, validated: bool = False):
        """`validated=True`: the caller built idx on the host and already checked its range (no device sync here)."""
        feats = self._dev(feats, torch.float32)
        if not validated and not (isinstance(idx, torch.Tensor) and idx.is_cuda):
            a = np.asarray(idx)
            if a.size and (a.min() < 0 or a.max() >= feats.shape[0]):
                raise ValueError("gather_windows: index out of range")
            validated = True
        idx = self._dev(idx, torch.int32)
        if idx.dim() != 2 or idx.shape[1] != 10 or feats.dim() != 2 or feats.shape[1] != 512:
            raise ValueError("gather_windows: feats [*,512], idx [nwin,10]")
        if not validated and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= feats.shape[0]):
            raise ValueError("gather_windows: index out of range")
        out = self._new(idx.shape[0], 10, 512)
        self._check(self.lib.avcer_gather_windows(self.ctx, _ptr(feats), _ptr(idx), int(idx.shape[0]), _ptr(out),
                                                  self._stream()))
        return out

    def dynamic_forward(self, windows, mode: int = MODE_DEFAULT):
        x = self._dev(windows, torch.float32)
        if x.dim() != 3 or tuple(x.shape[1:]) != (10, 512):
            raise ValueError(f"windows must be [N,10,512], got {tuple(x.shape)}")
        out = self._new(x.shape[0], 7)
        self._check(self.lib.avcer_dynamic_forward_mode(self.ctx, _ptr(x), int(x.shape[0]), int(mode), _ptr(out),
                                                        self._stream()))
        return out

    def audio_forward(self, wav, normalize: bool = True, mode: int = MODE_DEFAULT):
        x = self._dev(wav, torch.float32)
        if x.dim() != 2:
            raise ValueError(f"wav must be [N,T], got {tuple(x.shape)}")
        out = self._new(x.shape[0], self.audio_classes)
        self._check(self.lib.avcer_audio_forward(self.ctx, _ptr(x), int(x.shape[0]), int(x.shape[1]), int(normalize),
                                                 mode, _ptr(out), self._stream()))
        return out

    def audio_chunks(self, wav, starts, ends, window: int, padding: str = "mean"):
        wav = self._dev(wav, torch.float32)
        # the ranges are validated on the HOST when they come from the host (chunk_spans builds them there): a device-side
        # `int(starts.min())` is a synchronisation in front of every launch of the audio branch
        if not (torch.is_tensor(starts) and starts.is_cuda) and not (torch.is_tensor(ends) and ends.is_cuda):
            hs = np.asarray(starts.cpu() if torch.is_tensor(starts) else starts).reshape(-1).astype(np.int64)
            he = np.asarray(ends.cpu() if torch.is_tensor(ends) else ends).reshape(-1).astype(np.int64)
            n = int(hs.size)
            if wav.dim() != 1 or he.size != n:
                raise ValueError("audio_chunks: wav [L], starts/ends [n]")
            if n and (hs.min() < 0 or he.max() > wav.numel() or (he < hs).any() or (he - hs).max() > window):
                raise ValueError("audio_chunks: sample ranges out of bounds")
            if padding == "repeat" and n and (he - hs).min() == 0:
                raise ZeroDivisionError("integer division or modulo by zero")  # data/utils.py:66 on an empty chunk
            starts = torch.from_numpy(hs.astype(np.int32)).to(self.device, non_blocking=True)
            ends = torch.from_numpy(he.astype(np.int32)).to(self.device, non_blocking=True)
        else:
            starts = self._dev(starts, torch.int32)
            ends = self._dev(ends, torch.int32)
            n = int(starts.numel())
            if wav.dim() != 1 or ends.numel() != n:
                raise ValueError("audio_chunks: wav [L], starts/ends [n]")
            if n and (int(starts.min()) < 0 or int(ends.max()) > wav.numel() or bool((ends < starts).any())
                      or int((ends - starts).max()) > window):
                raise ValueError("audio_chunks: sample ranges out of bounds")
            if padding == "repeat" and n and int((ends - starts).min()) == 0:
                raise ZeroDivisionError("integer division or modulo by zero")  # data/utils.py:66 on an empty chunk
        out = self._new(n, window)
        self._check(self.lib.avcer_audio_chunks(self.ctx, _ptr(wav), _ptr(starts), _ptr(ends), n, int(window),
                                                PAD_MODES[padding], _ptr(out), self._stream()))
        return out

    def audio_frame_mean(self, win_logits, frame_lo, frame_hi, n_frames: int):
        x = self._dev(win_logits, torch.float32)
        lo, hi = self._dev(frame_lo, torch.int32), self._dev(frame_hi, torch.int32)
        out = self._new(n_frames, x.shape[1])
        cnt = self._new(n_frames, dtype=torch.int32)
        self._check(self.lib.avcer_audio_frame_mean(self.ctx, _ptr(x), _ptr(lo), _ptr(hi), int(x.shape[0]),
                                                    int(x.shape[1]), int(n_frames), _ptr(out), _ptr(cnt), self._stream()))
        return out, cnt

    def face_decode(self, loc, conf, landms, priors, image_size, variance=(0.1, 0.2)):
        """RetinaFace head outputs [P,4], [P,2], [P,10] + priors [P,4] -> dets [P,15] in pixels (before filtering)."""
        loc, conf = self._dev(loc, torch.float32), self._dev(conf, torch.float32)
        landms, priors = self._dev(landms, torch.float32), self._dev(priors, torch.float32)
        p = int(priors.shape[0])
        if tuple(loc.shape) != (p, 4) or tuple(conf.shape) != (p, 2) or tuple(landms.shape) != (p, 10) or priors.shape[1] != 4:
            raise ValueError("face_decode: loc [P,4], conf [P,2], landms [P,10], priors [P,4]")
        dets = self._new(p, 15)
        self._check(self.lib.avcer_face_decode(self.ctx, _ptr(loc), _ptr(conf), _ptr(landms), _ptr(priors), p,
                                               int(image_size[0]), int(image_size[1]), float(variance[0]),
                                               float(variance[1]), _ptr(dets), self._stream()))
        return dets

    def face_decode_batch(self, loc, conf, landms, priors, image_size, variance=(0.1, 0.2)):
        """The same for T frames in one launch: loc [T,P,4], conf [T,P,2], landms [T,P,10] + priors [P,4] -> dets [T,P,15]."""
        loc, conf = self._dev(loc, torch.float32), self._dev(conf, torch.float32)
        landms, priors = self._dev(landms, torch.float32), self._dev(priors, torch.float32)
        t, p = int(loc.shape[0]), int(priors.shape[0])
        if tuple(loc.shape) != (t, p, 4) or tuple(conf.shape) != (t, p, 2) or tuple(landms.shape) != (t, p, 10) or priors.shape[1] != 4:
            raise ValueError("face_decode_batch: loc [T,P,4], conf [T,P,2], landms [T,P,10], priors [P,4]")
        dets = self._new(t, p, 15)
        self._check(self.lib.avcer_face_decode_batch(self.ctx, _ptr(loc), _ptr(conf), _ptr(landms), _ptr(priors), t, p,
                                                     int(image_size[0]), int(image_size[1]), float(variance[0]),
                                                     float(variance[1]), _ptr(dets), self._stream()))
        return dets

    def face_nms(self, dets, conf_thresh: float = 0.02, nms_thresh: float = 0.4, nms_top_k: int = 5000, top_k: int = 750,
                 threshold: float = 0.8):
        """dets [T,P,15] (face_decode output per frame) -> (rows [T,top_k,15], counts [T]) after the reference's confidence
        floor, NMS, top-k and final threshold; only counts[t] rows of frame t are meaningful."""
        d = self._dev(dets, torch.float32)
        if d.dim() == 2:
            d = d[None]
        if d.dim() != 3 or d.shape[2] != 15:
            raise ValueError("face_nms: dets [T,P,15]")
        t, p = int(d.shape[0]), int(d.shape[1])
        out = self._new(t, top_k, 15)
        cnt = self._new(t, dtype=torch.int32)
        self._check(self.lib.avcer_face_nms(self.ctx, _ptr(d), t, p, float(conf_thresh), float(nms_thresh), int(nms_top_k),
                                            int(top_k), float(threshold), _ptr(out), _ptr(cnt), self._stream()))
        return out, cnt

    def track_faces(self, dets_per_frame, frame_w: int, frame_h: int, iou_threshold: float = 0.4, minimum_face_size: float = 0.0):
        """The tracker + crop-rectangle loop of `VideoPredictor.process` for a whole video in one native HOST call
        (csrc/track.hip): per-frame detection arrays [k, >= 4] -> records int64 [n, 6] = frame, track directory, x0, y0, x1, y1.
        Raises ValueError where the reference raises (zero-area detection, empty crop)."""
        counts = np.fromiter((len(d) for d in dets_per_frame), dtype=np.int32, count=len(dets_per_frame))
        total = int(counts.sum())
        boxes = np.zeros((max(total, 1), 4), dtype=np.float32)
        if total:
            np.concatenate([np.asarray(d, dtype=np.float32).reshape(len(d), -1)[:, :4] for d in dets_per_frame if len(d)], out=boxes[:total])
        rec = np.empty((max(total, 1), 6), dtype=np.int64)
        n = C.c_int64(0)
        rc = self.lib.avcer_track_faces(self.ctx, boxes.ctypes.data_as(C.c_void_p), 4, counts.ctypes.data_as(C.c_void_p), len(counts),
                                        int(frame_w), int(frame_h), float(iou_threshold), float(minimum_face_size),
                                        rec.ctypes.data_as(C.c_void_p), C.byref(n))
        if rc == -1:
            raise ValueError(self.lib.avcer_last_error(self.ctx).decode("utf-8", "replace"))
        self._check(rc)
        return rec[:int(n.value)]

    def crop_tiles(self, frames_u8, rects, bgr: bool = True):
        """frames u8 [T,H,W,3] + rects i32 [n,5] (frame, x0, y0, x1, y1; validated by the caller) -> RGB tiles [n,224,224,3]."""
        x = self._dev(frames_u8, torch.uint8)
        r = self._dev(rects, torch.int32)
        if x.dim() != 4 or x.shape[-1] != 3 or r.dim() != 2 or r.shape[1] != 5:
            raise ValueError("crop_tiles: frames [T,H,W,3] uint8, rects [n,5] int32")
        n = int(r.shape[0])
        tiles = self._new(n, 224, 224, 3, dtype=torch.uint8)
        self._check(self.lib.avcer_crop_tiles(self.ctx, _ptr(x), int(x.shape[0]), int(x.shape[1]), int(x.shape[2]),
                                              _ptr(r), n, 1 if bgr else 0, _ptr(tiles), self._stream()))
        return tiles

    def fuse(self, stat, dyn_logits, aud_mean, n_aud: int, weights_1=None, weights_2=(1, 1, 1),
             ce_weights_type: bool = False, ce_mask: bool = True):
        stat = self._dev(stat, torch.float32)
        dyn = self._dev(dyn_logits, torch.float32)
        aud = self._dev(aud_mean, torch.float32)
        n = int(stat.shape[0])
        if tuple(stat.shape) != (n, 7) or tuple(dyn.shape) != (n, 7) or aud.dim() != 2 or aud.shape[1] < 7:
            raise ValueError("fuse: stat/dyn [n,7], aud [n_aud,>=7]")
        if not (1 <= n_aud <= aud.shape[0]):
            raise ValueError("fuse: n_aud out of range")
        w1 = (C.c_double * 21)(*np.asarray(weights_1, dtype=np.float64).reshape(21)) if weights_1 else None
        w2 = (C.c_double * 3)(*[float(v) for v in weights_2])
        prob = self._new(4, n, 7, dtype=torch.float64)
        am = self._new(4, n, dtype=torch.int32)
        self._check(self.lib.avcer_fuse(self.ctx, _ptr(stat), _ptr(dyn), _ptr(aud), n, int(n_aud), int(aud.shape[1]),
                                        w1, w2, int(bool(ce_weights_type)), int(bool(ce_mask)), _ptr(prob), _ptr(am),
                                        self._stream()))
        return prob, am

    def conv_gemm(self, desc: ConvDesc, dtype: int, x, w, scale, bias, residual, y):
        self._check(self.lib.avcer_conv_gemm(self.ctx, C.byref(desc), dtype, _ptr(x), _ptr(w), _ptr(scale), _ptr(bias),
                                             _ptr(residual), _ptr(y), self._stream()))

    def profile_enable(self, on: bool = True):
        self._check(self.lib.avcer_profile_enable(self.ctx, int(on)))

    def profile_read(self):
        """(summed conv_gemm kernel time in ms, number of launches) since the last read; synchronises."""
        ms, n = C.c_double(0.0), C.c_int64(0)
        self._check(self.lib.avcer_profile_read(self.ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    FAMILIES = ("conv_gemm_kernel", "conv_gemm_wd_kernel", "bneck_kernel", "bneck_tail2_kernel", "stem_pool_kernel",
                "conv_gemm_skinny_kernel")  # AVCER_FAM_*

    def profile_read_families(self):
        """Per kernel family since profile_enable / the last read: {name: (event ms, launches, algorithmic FLOPs, compulsory
        HBM bytes)}; synchronises.  Use instead of profile_read."""
        n = len(self.FAMILIES)
        ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        la = (C.c_int64 * n)()
        self._check(self.lib.avcer_profile_read_families(self.ctx, n, ms, la, fl, by))
        return {name: (ms[i], int(la[i]), fl[i], by[i]) for i, name in enumerate(self.FAMILIES)}

    def profile_read_launches(self, max_n: int = 8192):
        """Launch by launch since profile_enable / the last read: list of dicts {family, ms, flops, bytes, m, n, k} in launch
        order; synchronises.  Use instead of profile_read / profile_read_families."""
        fam = np.zeros(max_n, np.int32)
        ms, fl, by = np.zeros(max_n), np.zeros(max_n), np.zeros(max_n)
        mnk = np.zeros((max_n, 3), np.int64)
        n = C.c_int64(0)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        self._check(self.lib.avcer_profile_read_launches(self.ctx, max_n, vp(fam), vp(ms), vp(fl), vp(by), vp(mnk), C.byref(n)))
        k = min(int(n.value), max_n)
        return [dict(family=self.FAMILIES[fam[i]], ms=float(ms[i]), flops=float(fl[i]), bytes=float(by[i]), m=int(mnk[i, 0]),
                     n=int(mnk[i, 1]), k=int(mnk[i, 2])) for i in range(k)]

    def debug_tap(self, name: str, numel: int, dtype=torch.float32):
        """Arm a one-shot tap; returns the destination tensor (filled by the next forward pass)."""
        dst = torch.zeros(numel, dtype=dtype, device=self.device)
        self._tap_keepalive = dst
        self._check(self.lib.avcer_debug_tap(self.ctx, name.encode(), _ptr(dst), dst.numel() * dst.element_size()))
        return dst

    def debug_tap_copied(self) -> int:
        return int(self.lib.avcer_debug_tap_copied(self.ctx))

    def split_weights(self, w):
        """f32 tensor (numel a multiple of 32) -> sp32 ACTIVATION layout (hi / lo fp16 per group of 32; int16 tensor of
        2 * numel entries): the A operand of conv_gemm dtypes 5 / 6.  Not valid for weights: use split_weight_rows."""
        w = self._dev(w, torch.float32)
        out = torch.empty(w.numel() * 2, dtype=torch.int16, device=self.device)
        self._check(self.lib.avcer_split_weights(self.ctx, _ptr(w), _ptr(out), w.numel(), self._stream()))
        return out

    def split_weight_rows(self, w):
        """f32 [N,K] weight matrix -> the split-fp16, row-permuted layout of conv_gemm dtypes 3-6 and the fused kernels
        (int16 tensor of 2*N*K entries)."""
        w = self._dev(w, torch.float32)
        if w.dim() != 2:
            raise ValueError("split_weight_rows: w [N,K]")
        out = torch.empty(w.numel() * 2 + SPLIT_TRAILER // 2, dtype=torch.int16, device=self.device)
        self._check(self.lib.avcer_split_weight_rows(self.ctx, _ptr(w), _ptr(out), int(w.shape[0]), int(w.shape[1]), self._stream()))
        return out

    def weight_frags(self, w):
        """f32 [N,K] weight matrix -> the fragment-order split layout of conv_gemm dtypes 7 / 8 (int16 tensor of 2*N*K entries)."""
        w = self._dev(w, torch.float32)
        rows = self.split_weight_rows(w)
        out = torch.empty_like(rows)
        self._check(self.lib.avcer_weight_frags(self.ctx, _ptr(rows), _ptr(out), int(w.shape[0]), int(w.shape[1]), self._stream()))
        return out

    def conv_gemm_dual(self, desc: ConvDesc, dtype: int, x, x2, w, scale, bias, residual, y):
        self._check(self.lib.avcer_conv_gemm_dual(self.ctx, C.byref(desc), dtype, _ptr(x), _ptr(x2), _ptr(w), _ptr(scale),
                                                  _ptr(bias), _ptr(residual), _ptr(y), self._stream()))

    def bneck_chain(self, planes: int, nb: int, h: int, w: int, t1, x, out, t1n, w2, b2, w3, b3, w1n=None, b1n=None, ds_cin: int = 0,
                    out_step: int = 1, w2_frags=None):
        """Kernel-level entry of the fused bottleneck chain (csrc/fused.hip); all tensors already on the device.
        out_step = 2: the last block of a stage, evaluated at the even positions only (out is the compact grid).
        w2_frags: `weight_frags` of the conv2 matrix: selects the spatial-tile form where it applies (planes 64, 55 x 55)."""
        self._check(self.lib.avcer_bneck_chain(self.ctx, planes, nb, h, w, _ptr(t1), _ptr(x), int(ds_cin), int(out_step), _ptr(out),
                                               _ptr(t1n), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), _ptr(w1n), _ptr(b1n),
                                               _ptr(w2_frags), self._stream()))

    def attention(self, qkv, out, n: int, s: int, heads: int, head_dim: int, scale: float, in_kind: int, out_kind: int):
        """Kernel-level entry of the attention kernel: qkv [n, s, 3 * heads * head_dim] -> out [n, s, heads * head_dim];
        storage kinds 0 = f32, 1 = bf16, 2 = sp32 (int16 tensor of twice the elements)."""
        self._check(self.lib.avcer_attention(self.ctx, _ptr(qkv), _ptr(out), n, s, heads, head_dim, float(scale), in_kind,
                                             out_kind, self._stream()))

    def measure_ceilings(self):
        """(f16 MFMA TFLOP/s of a register-only v_mfma_f32_16x16x32_f16 loop, TB/s of a 1 GiB streaming copy) measured on this GPU."""
        a, b = C.c_double(0.0), C.c_double(0.0)
        self._check(self.lib.avcer_measure_ceilings(self.ctx, C.byref(a), C.byref(b), self._stream()))
        return a.value, b.value

    def stem_pool(self, planes_hi_lo, w_split, scale, bias, n: int):
        """Kernel-level entry of the fused stem (csrc/fused.hip): planes int16 [2, n, 230, 230, 4] (fp16 hi plane, lo plane)
        -> sp32 [n, 55, 55, 64] as int16 [n, 55, 55, 128]."""
        planes = self._dev(planes_hi_lo, torch.int16)
        if tuple(planes.shape) != (2, n, 230, 230, 4):
            raise ValueError("stem_pool: planes [2, n, 230, 230, 4] int16 (fp16 bits)")
        y = torch.empty(n, 55, 55, 128, dtype=torch.int16, device=self.device)
        self._check(self.lib.avcer_stem_pool(self.ctx, _ptr(planes), n * 230 * 230 * 4 * 2, _ptr(w_split), _ptr(scale), _ptr(bias),
                                             _ptr(y), n, self._stream()))
        return y

    def stem_pool_u8(self, frames_u8, w_split, scale, shifts9):
        """The fused stem fed with u8 frames [n, h, w, 3] RGB (preprocessing inside, raw pixels as exact fp16 operands);
        shifts9 f32 [9, 64] from packing.stem_border_shifts.  -> sp32 [n, 55, 55, 64] as int16 [n, 55, 55, 128]."""
        fr = self._dev(frames_u8, torch.uint8)
        n, h, w = int(fr.shape[0]), int(fr.shape[1]), int(fr.shape[2])
        y = torch.empty(n, 55, 55, 128, dtype=torch.int16, device=self.device)
        self._check(self.lib.avcer_stem_pool_u8(self.ctx, _ptr(fr), n, h, w, _ptr(w_split), _ptr(scale), _ptr(shifts9), _ptr(y),
                                                self._stream()))
        return y

    def gemm_stats(self, reset: bool = True):
        n, f = C.c_int64(0), C.c_double(0.0)
        self._check(self.lib.avcer_gemm_stats(self.ctx, C.byref(n), C.byref(f), int(reset)))
        return n.value, f.value
