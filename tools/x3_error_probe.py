#!/usr/bin/env python3
"""Which rounding owns the error of the split-fp16 ("x3") contraction?  (fp16 pairs since round 4.)

For y = x @ w^T with post-ReLU-like x >= 0 and zero-mean w, against the float64 product of the SAME f32 inputs:

    total        conv_gemm dtype 3 (f32 in, split on the fly, f32 out)          - exact
    operands     float64 of (xh wh + xh wl + xl wh) on the host-side splits     - exact   (split rounding + dropped xl wl)
    dropped      float64 of  xl wl                                                          (the fourth product alone)
    accumulate   conv_gemm dtype 3 - float64 of (xh wh + xh wl + xl wh)                     (MFMA / f32 accumulation only)

each as rms relative to rms(y), plus the SIGNED mean of the accumulation error along sign(y) (a truncating adder would
show as a negative bias that grows with K).  The f32 MFMA path (dtype 0) is printed next to it as the noise floor.

    python tools/x3_error_probe.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avcer_amd._lib import ConvDesc  # noqa: E402
from avcer_amd.engine import Engine  # noqa: E402


def split(a):
    """f32 -> (hi, lo) fp16 pair as float32 arrays (numpy rounds to nearest even and keeps subnormals, like the GPU)."""
    a = a.astype(np.float32)
    h = a.astype(np.float16).astype(np.float32)
    return h, (a - h).astype(np.float16).astype(np.float32)


def weight_scale(w):
    """The power of two avcer_split_weight_rows multiplies a matrix by: largest magnitude -> [2^14, 2^15)."""
    return np.float32(2.0 ** (14 - int(np.floor(np.log2(np.abs(w).max())))))


def linear_desc(m, k, n):
    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.out_h, d.out_w = m, 1, 1, 1, 1
    d.cin, d.kh, d.kw = k, 1, 1
    d.stride_h = d.stride_w = d.dil_h = d.dil_w = 1
    d.x_stride_b = d.x_stride_h = d.x_stride_w = k
    d.n, d.y_ld, d.r_ld = n, n, n
    return d


def main():
    eng = Engine(0)
    rng = np.random.default_rng(0)
    m, n = 2048, 128
    print(f"{'K':>6} {'xscale':<7} {'total':>10} {'operands':>10} {'dropped':>10} {'accum rms':>10} {'accum bias':>11} {'f32 MFMA':>10}")
    for k, xs in ((64, 1.0), (256, 1.0), (1024, 1.0), (2304, 1.0), (4608, 1.0), (1024, 0.02), (1024, 300.0)):
        x = (np.maximum(rng.standard_normal((m, k)), 0) * xs).astype(np.float32)  # xs: activation scale (fp16 range probe)
        w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        exact = x.astype(np.float64) @ w.astype(np.float64).T
        xh, xl = split(x)
        ws = weight_scale(w)
        wh, wl = split(w * ws)
        f = np.float64
        x3 = (xh.astype(f) @ wh.astype(f).T + xh.astype(f) @ wl.astype(f).T + xl.astype(f) @ wh.astype(f).T) / f(ws)
        dropped = xl.astype(f) @ wl.astype(f).T / f(ws)
        xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
        y3, y0 = torch.empty(m, n, device="cuda"), torch.empty(m, n, device="cuda")
        eng.conv_gemm(linear_desc(m, k, n), 3, xd, eng.split_weight_rows(wd), None, None, None, y3)
        eng.conv_gemm(linear_desc(m, k, n), 0, xd, wd, None, None, None, y0)
        torch.cuda.synchronize()
        g3, g0 = y3.cpu().numpy().astype(f), y0.cpu().numpy().astype(f)
        s = np.sqrt((exact ** 2).mean())
        rms = lambda e: np.sqrt((e ** 2).mean()) / s  # noqa: E731
        acc = g3 - x3
        bias = (acc * np.sign(exact)).mean() / s
        print(f"{k:6d} x{xs:<6g} {rms(g3 - exact):10.2e} {rms(x3 - exact):10.2e} {rms(dropped):10.2e} {rms(acc):10.2e} {bias:11.2e} "
              f"{rms(g0 - exact):10.2e}")


if __name__ == "__main__":
    main()
