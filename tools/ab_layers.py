#!/usr/bin/env python3
"""A/B of the two forms of the split-fp16 contraction on the hot path's layer shapes, interleaved in ONE process
(rule: never compare timings taken in different processes or on different boxes):

    staged   avcer_conv_gemm dtype 5 / 6  (A and W tiles through LDS, conv_gemm_kernel<3, *, *>)
    direct   avcer_conv_gemm dtype 7 / 8  (W fragments straight to registers, conv_gemm_wd_kernel, 128 x 256 tiles and
             112 x 256 tiles: avcer_conv_desc.tile_m)

Inputs are real sp32 encodings of random activations (post-ReLU-like), not reinterpreted f32 bits.  Every arm gets the same
warm-up; each round times `iters` back-to-back launches per arm; the median over rounds is printed.

    python tools/ab_layers.py [--frames 2048] [--chunks 128] [--rounds 5] [--iters 10] [--only substring]
"""
from __future__ import annotations

import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.layer_bench import audio_layers, static_layers  # noqa: E402
from avcer_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--chunks", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--f32out", action="store_true", help="f32 output (dtype 6 / 8) instead of sp32 (5 / 7)")
    ap.add_argument("--act", type=int, default=-1, help="override the layers' activation (0 none, 1 relu, 2 gelu, 3 gelu with the short erf)")
    ap.add_argument("--skinny", action="store_true", help="arms of the skinny form (dtype 9 / 10; tiles of 16 / 32 / 64 positions) where M <= 4096")
    ap.add_argument("--lib", default="", help="a one-off experiment build of libavcer_hip.so to load instead of the in-tree one")
    a = ap.parse_args()
    if a.lib:
        from avcer_amd import _lib
        _lib.LIB = os.path.abspath(a.lib)
    eng = Engine(0)
    dev = eng.device
    layers = [L for L in static_layers(a.frames) + audio_layers(a.chunks, 32000) if a.only in L["name"]]
    tot = {"staged": 0.0, "best": 0.0}
    print(f"{'layer':38s} {'x':>3s} {'staged us':>10s} {'direct128':>10s} {'direct112':>10s} {'library':>10s} {'best/staged':>11s}")
    for L in layers:
        d = L["d"]
        if a.act >= 0:
            d.act = a.act
        m, k, n = d.batch * d.out_h * d.out_w, d.kh * d.kw * d.cin, d.n
        g = L.get("groups", 1)
        if d.x_stride_w % 32 or d.cin % 32 or g != 1 or n % 256 or (k // 32) % 2 or d.y_ld % 32:
            continue
        nel = max(L["in_elems"], d.x_stride_b * d.batch) + 64
        nel += (-nel) % 32
        if nel * 4 >= 0xF0000000:
            print(f"{L['name']:38s} skipped: input beyond the 4 GiB descriptor range at this batch")
            continue
        x = eng.split_weights(torch.relu(torch.randn(nel, device=dev)))         # sp32 encoding of post-ReLU activations
        w = torch.randn(n, k, device=dev) / k ** 0.5
        rows, frags = eng.split_weight_rows(w), eng.weight_frags(w)
        ylen = m * n + 64
        y = torch.empty(ylen * 2, device=dev, dtype=torch.int16)
        res = eng.split_weights(torch.randn(ylen + (-ylen) % 32, device=dev)) if L["res"] else None
        sc, bi = torch.ones(n, device=dev), torch.zeros(n, device=dev)
        ds, dd = (6, 8) if a.f32out else (5, 7)
        if a.f32out:
            y = torch.empty(ylen, device=dev)
            res = torch.randn(ylen, device=dev) if L["res"] else None
        arms = [("staged", ds, rows, 0), ("direct128", dd, frags, 128), ("direct112", dd, frags, 112), ("library", dd, frags, 0)]
        if a.skinny:
            if m > 4096:
                continue
            arms = [("staged", ds, rows, 0), ("library", dd, frags, 0)] + [(f"skinny{t}", dd + 2, frags, t) for t in (16, 32, 64)]

        def launch(dt, wt, tm, cnt):
            d.tile_m = tm
            for _ in range(cnt):
                eng.conv_gemm(d, dt, x, wt, sc, bi, res, y)

        for _, dt, wt, tn in arms:
            launch(dt, wt, tn, 3)
        torch.cuda.synchronize()
        times = {name: [] for name, *_ in arms}
        for _ in range(a.rounds):
            for name, dt, wt, tn in arms:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch(dt, wt, tn, a.iters)
                e1.record()
                torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1) / a.iters * 1e3)
        med = {name: statistics.median(v) for name, v in times.items()}
        best = min(med.values())
        tot["staged"] += med["staged"] * L["count"]
        tot["best"] += best * L["count"]
        if a.skinny:
            print(f"{L['name']:38s} M={m:5d} K={k:5d} N={n:5d}  " + "  ".join(f"{nm} {v:7.1f}" for nm, v in med.items()))
        else:
            print(f"{L['name']:38s} {L['count']:3d} {med['staged']:10.1f} {med['direct128']:10.1f} {med['direct112']:10.1f} "
                  f"{med['library']:10.1f} {best / med['staged']:11.3f}")
        d.tile_m = 0
        del x, w, rows, frags, y, res
    print(f"sum over the layers above (x count): staged {tot['staged'] / 1e3:.2f} ms, best arm per layer {tot['best'] / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
