"""ctypes binding of libavcer_hip.so (include/avcer_hip.h).  There is no CPU fallback: if the library is
missing or fails to load, importing a model class raises."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB, source_hash

c_ctx = C.c_void_p
c_stream = C.c_void_p


class ConvDesc(C.Structure):
    """struct avcer_conv_desc"""
    _fields_ = [
        ("batch", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
        ("out_h", C.c_int32), ("out_w", C.c_int32),
        ("cin", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
        ("stride_h", C.c_int32), ("stride_w", C.c_int32), ("pad_h", C.c_int32), ("pad_w", C.c_int32),
        ("dil_h", C.c_int32), ("dil_w", C.c_int32),
        ("x_stride_b", C.c_int64), ("x_stride_h", C.c_int64), ("x_stride_w", C.c_int64),
        ("x_coff", C.c_int32), ("n", C.c_int32),
        ("y_ld", C.c_int64), ("y_coff", C.c_int32),
        ("r_ld", C.c_int64), ("r_coff", C.c_int32),
        ("act", C.c_int32), ("res_after_act", C.c_int32), ("groups", C.c_int32),
        ("x2_cin", C.c_int32), ("x2_coff", C.c_int32), ("x2_stride", C.c_int32),
        ("x2_stride_b", C.c_int64), ("x2_stride_h", C.c_int64), ("x2_stride_w", C.c_int64),
        ("tile_n", C.c_int32),
        ("r_sub", C.c_int32), ("r_h", C.c_int32), ("r_w", C.c_int32),
        ("tile_m", C.c_int32),
    ]


ABI_VERSION = 4          # include/avcer_hip.h AVCER_ABI_VERSION: struct layouts, argument lists and buffer sizes below
SPLIT_TRAILER = 256      # include/avcer_hip.h AVCER_SPLIT_TRAILER: bytes behind a split weight matrix (its scale)

# name -> (restype, argtypes); exactly the symbols include/avcer_hip.h declares
SIGNATURES = {
    "avcer_abi_version": (C.c_int, []),
    "avcer_source_hash": (C.c_char_p, []),
    "avcer_ctx_create": (C.c_int, [C.c_int, C.POINTER(c_ctx)]),
    "avcer_ctx_destroy": (None, [c_ctx]),
    "avcer_last_error": (C.c_char_p, [c_ctx]),
    "avcer_x3_overflow_count": (C.c_int, [c_ctx, C.c_int, C.POINTER(C.c_int64), c_stream]),
    "avcer_load_static": (C.c_int, [c_ctx, C.c_void_p, C.c_size_t]),
    "avcer_load_dynamic": (C.c_int, [c_ctx, C.c_void_p, C.c_size_t]),
    "avcer_load_audio": (C.c_int, [c_ctx, C.c_void_p, C.c_size_t]),
    "avcer_static_forward": (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, c_stream]),
    "avcer_set_static_batch": (C.c_int, [c_ctx, C.c_int]),
    "avcer_set_static_back_batch": (C.c_int, [c_ctx, C.c_int]),
    "avcer_set_static_lanes": (C.c_int, [c_ctx, C.c_int]),
    "avcer_set_static_lane_range": (C.c_int, [c_ctx, C.c_int, C.c_int]),
    "avcer_static_forward_nchw": (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            c_stream]),
    "avcer_gather_windows": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, c_stream]),
    "avcer_dynamic_forward": (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_void_p, c_stream]),
    "avcer_dynamic_forward_mode": (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_stream]),
    "avcer_audio_forward": (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, c_stream]),
    "avcer_audio_num_classes": (C.c_int, [c_ctx]),
    "avcer_audio_chunks": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     c_stream]),
    "avcer_audio_frame_mean": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, c_stream]),
    "avcer_load_face": (C.c_int, [c_ctx, C.c_void_p, C.c_size_t]),
    "avcer_face_num_priors": (C.c_int, [C.c_int, C.c_int]),
    "avcer_face_forward": (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, c_stream]),
    "avcer_face_decode": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_float, C.c_float, C.c_void_p, c_stream]),
    "avcer_face_decode_batch": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_float, C.c_float, C.c_void_p, c_stream]),
    "avcer_track_faces": (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                    C.c_void_p, C.POINTER(C.c_int64)]),
    "avcer_lsap": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "avcer_crop_tiles": (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, c_stream]),
    "avcer_fuse": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                             C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, c_stream]),
    "avcer_conv_gemm": (C.c_int, [c_ctx, C.POINTER(ConvDesc), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, c_stream]),
    "avcer_split_weights": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_size_t, c_stream]),
    "avcer_split_weight_rows": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_int, c_stream]),
    "avcer_weight_frags": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_int, c_stream]),
    "avcer_conv_gemm_dual": (C.c_int, [c_ctx, C.POINTER(ConvDesc), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, c_stream]),
    "avcer_face_nms": (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float,
                                 C.c_void_p, C.c_void_p, c_stream]),
    "avcer_bneck_chain": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 9 +
                          [c_stream]),
    "avcer_stem_pool": (C.c_int, [c_ctx, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_stream]),
    "avcer_stem_pool_u8": (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     c_stream]),
    "avcer_attention": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                  c_stream]),
    "avcer_measure_ceilings": (C.c_int, [c_ctx, C.POINTER(C.c_double), C.POINTER(C.c_double), c_stream]),
    "avcer_gemm_stats": (C.c_int, [c_ctx, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_int]),
    "avcer_profile_enable": (C.c_int, [c_ctx, C.c_int]),
    "avcer_profile_read": (C.c_int, [c_ctx, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "avcer_profile_read_families": (C.c_int, [c_ctx, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                              C.POINTER(C.c_double)]),
    "avcer_profile_read_launches": (C.c_int, [c_ctx, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.POINTER(C.c_int64)]),
    "avcer_debug_tap": (C.c_int, [c_ctx, C.c_char_p, C.c_void_p, C.c_size_t]),
    "avcer_debug_tap_copied": (C.c_int64, [c_ctx]),
}

_lib = None
_PRODUCT_LIB = LIB  # `LIB` may be overridden by lab tools; the stamp check applies to the product path only


def load() -> C.CDLL:
    """dlopen the in-tree library (built by `python -m avcer_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(
                f"{LIB} is missing: run `python -m avcer_amd.build` (hipcc, gfx950). "
                "avcer_amd has no CPU or PyTorch fallback for the hot path.")
        import torch  # noqa: F401  -- loads the process's one HIP runtime (libamdhip64.so.7) before the dlopen

        lib = C.CDLL(LIB, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.avcer_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB}: ABI version {lib.avcer_abi_version()}, this binding is written for {ABI_VERSION} "
                               "(include/avcer_hip.h AVCER_ABI_VERSION): rebuild with `python -m avcer_amd.build`")
        check_source_hash(lib, LIB)
        _lib = lib
    return _lib


def check_source_hash(lib, path: str, tree_hash: str | None = None) -> str:
    """The library embeds the hash of the sources it was compiled from (avcer_source_hash(), written by build.py); a binary
    built from OTHER sources than the tree's -- a stale .so behind an unchanged ABI number -- is refused.  One-off lab builds
    loaded through an overridden `_lib.LIB` (tools/: patched copies, their hash ends in "+lab...") are exempt."""
    have = lib.avcer_source_hash().decode()
    want = tree_hash if tree_hash is not None else source_hash()
    if "+lab" in have or os.path.abspath(path) != os.path.abspath(_PRODUCT_LIB):
        return have
    if have != want:
        raise RuntimeError(f"{path} was built from sources with hash {have}, the tree's csrc/ + include/ hash is {want}: "
                           "stale binary -- rebuild with `python -m avcer_amd.build`")
    return have


class AvcerError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libavcer_hip error {code}: {msg}")
        self.code = code
