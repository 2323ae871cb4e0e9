"""Host-side view of the "sp32" storage of AVCER_MODE_F16X3 (csrc/split_dev.h): per aligned group of 32 channels, 32 fp16 hi
values then 32 fp16 lo values, x = hi + lo; 4 bytes per element.  Used by tests and tools to build kernel inputs and to read
debug taps; the product path never converts on the host."""
from __future__ import annotations

import torch

from ._lib import SPLIT_TRAILER


def to_sp32(x: torch.Tensor) -> torch.Tensor:
    """f32 [..., C] (C a multiple of 32) -> sp32 storage as int16 [..., 2C] (round to nearest even, like the kernels)."""
    x = x.float()
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    c = x.shape[-1]
    h = hi.contiguous().view(torch.int16).reshape(*x.shape[:-1], c // 32, 32)
    l = lo.contiguous().view(torch.int16).reshape(*x.shape[:-1], c // 32, 32)
    return torch.cat([h, l], dim=-1).reshape(*x.shape[:-1], 2 * c).contiguous()


def from_sp32(s: torch.Tensor) -> torch.Tensor:
    """int16 [..., 2C] sp32 storage -> f32 [..., C]."""
    g = s.contiguous().reshape(*s.shape[:-1], s.shape[-1] // 64, 64)
    hi = g[..., :32].contiguous().view(torch.float16).float()
    lo = g[..., 32:].contiguous().view(torch.float16).float()
    return (hi + lo).reshape(*s.shape[:-1], s.shape[-1] // 2)


def raw_to_f32(raw_i16: torch.Tensor, shape) -> torch.Tensor:
    """A debug tap of an sp32 activation (flat int16) -> f32 tensor of `shape` ([..., C])."""
    return from_sp32(raw_i16.reshape(*shape[:-1], 2 * shape[-1])).reshape(shape)


def split_weight_mul(buf_i16: torch.Tensor, n: int, k: int) -> float:
    """The accumulator multiplier (a power of two) stored behind a split weight matrix of avcer_split_weight_rows /
    avcer_weight_frags: stored weights = w / mul."""
    assert buf_i16.numel() == 2 * n * k + SPLIT_TRAILER // 2
    return float(buf_i16[2 * n * k: 2 * n * k + 2].cpu().view(torch.float32)[0])
