#!/usr/bin/env python3
"""Per-layer micro-benchmark of the contraction kernel on the shapes of the AVCER hot path (GPU only).

    python tools/layer_bench.py [--dtype f32|bf16] [--frames 256] [--chunks 128] [--iters 5]

Prints one line per distinct (shape) with time, TFLOP/s and compulsory GB/s, and the totals per model.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avcer_amd._lib import ConvDesc  # noqa: E402
from avcer_amd.engine import Engine  # noqa: E402


def desc(**kw):
    d = ConvDesc()
    base = dict(batch=1, in_h=1, in_w=1, out_h=1, out_w=1, cin=32, kh=1, kw=1, stride_h=1, stride_w=1, pad_h=0, pad_w=0,
                dil_h=1, dil_w=1, x_stride_b=0, x_stride_h=0, x_stride_w=0, x_coff=0, n=64, y_ld=64, y_coff=0, r_ld=64,
                r_coff=0, act=0, res_after_act=0, groups=0, x2_cin=0, x2_coff=0, x2_stride=0, x2_stride_b=0, x2_stride_h=0,
                x2_stride_w=0, tile_n=0)
    base.update(kw)
    for k, v in base.items():
        setattr(d, k, int(v))
    return d


def conv2d(nb, h, c, k, stride, pad, n, name, count=1, res=False):
    oh = (h + 2 * pad - k) // stride + 1
    return dict(name=name, count=count, res=res, in_elems=nb * h * h * c, out_elems=nb * oh * oh * n,
                d=desc(batch=nb, in_h=h, in_w=h, out_h=oh, out_w=oh, cin=c, kh=k, kw=k, stride_h=stride, stride_w=stride,
                       pad_h=pad, pad_w=pad, x_stride_b=h * h * c, x_stride_h=h * c, x_stride_w=c, n=n, y_ld=n, r_ld=n, act=1))


def linear(m, k, n, name, count=1, res=False):
    return dict(name=name, count=count, res=res, in_elems=m * k, out_elems=m * n,
                d=desc(batch=m, cin=k, x_stride_b=k, x_stride_h=k, x_stride_w=k, n=n, y_ld=n, r_ld=n))


def conv1d(nb, length, c, k, stride, n, name, count=1):
    ol = (length - k) // stride + 1
    return dict(name=name, count=count, res=False, in_elems=nb * length * c, out_elems=nb * ol * n,
                d=desc(batch=nb, in_h=length, in_w=1, out_h=ol, out_w=1, cin=c, kh=k, kw=1, stride_h=stride,
                       x_stride_b=length * c, x_stride_h=c, x_stride_w=c, n=n, y_ld=n, r_ld=n))


def static_layers(nb):
    L = [dict(name="stem 8x(8x4)->64 s2", count=1, res=False, in_elems=nb * 230 * 230 * 4, out_elems=nb * 112 * 112 * 64,
              d=desc(batch=nb, in_h=230, in_w=230, out_h=112, out_w=112, cin=32, kh=8, kw=1, stride_h=2, stride_w=2,
                     x_stride_b=230 * 230 * 4, x_stride_h=230 * 4, x_stride_w=4, n=64, y_ld=64, r_ld=64, act=1))]
    h, cin = 55, 64
    for li, (p, blocks, s) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), 1):
        oh = (h - 1) // s + 1
        L.append(conv2d(nb, h, cin, 1, s, 0, p, f"l{li}.0.c1 {cin}->{p} s{s}"))
        L.append(conv2d(nb, h, cin, 1, s, 0, 4 * p, f"l{li}.0.ds {cin}->{4*p} s{s}"))
        L.append(conv2d(nb, oh, p, 3, 1, 1, p, f"l{li}.x.c2 3x3 {p}", blocks))
        L.append(conv2d(nb, oh, p, 1, 1, 0, 4 * p, f"l{li}.x.c3 {p}->{4*p}+res", blocks, res=True))
        L.append(conv2d(nb, oh, 4 * p, 1, 1, 0, p, f"l{li}.x.c1 {4*p}->{p}", blocks - 1))
        h, cin = oh, 4 * p
    L.append(linear(nb, 2048, 512, "fc1"))
    return L


def audio_layers(nb, t):
    ck, cs = (10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)
    ln = [t]
    for k, s in zip(ck, cs):
        ln.append((ln[-1] - k) // s + 1)
    S = ln[7]
    L = [conv1d(nb, ln[i], 512, ck[i], cs[i], 512, f"fe{i} k{ck[i]} s{cs[i]} L{ln[i]}->{ln[i+1]}") for i in range(1, 7)]
    r = nb * S
    L.append(linear(r, 512, 1024, "feature projection"))
    pos = dict(name="pos-conv 16 groups (k128, 64->64)", count=1, res=True, in_elems=r * 1024, out_elems=r * 1024, groups=16,
               d=desc(batch=nb, in_h=S, in_w=1, out_h=S, out_w=1, cin=64, kh=128, kw=1, pad_h=64, x_stride_b=S * 1024,
                      x_stride_h=1024, x_stride_w=1024, n=64, y_ld=1024, r_ld=1024, act=2, res_after_act=1, groups=16))
    L.append(pos)
    L += [linear(r, 1024, 3072, "qkv 1024->3072", 14), linear(r, 1024, 1024, "out-proj / tl ffn 1024->1024", 18, res=True),
          linear(r, 1024, 4096, "ffn1 1024->4096", 12), linear(r, 4096, 1024, "ffn2 4096->1024", 12, res=True)]
    L.append(dict(name="td0 conv k5 s3 dil2", count=1, res=False, in_elems=r * 1024, out_elems=nb * ((S - 9) // 3 + 1) * 1024,
                  d=desc(batch=nb, in_h=S, in_w=1, out_h=(S - 9) // 3 + 1, out_w=1, cin=1024, kh=5, kw=1, stride_h=3, dil_h=2,
                         x_stride_b=S * 1024, x_stride_h=1024, x_stride_w=1024, n=1024, y_ld=1024, r_ld=1024)))
    return L


def run(engine, layers, dtype, iters, title, tile_n=0):
    tin = torch.bfloat16 if dtype in (1, 2) else torch.float32
    es = 2 if dtype in (1, 2) else 4
    tot_ms = tot_fl = 0.0
    print(f"--- {title}")
    for L in layers:
        d = L["d"]
        d.tile_n = tile_n if ((tile_n != 128 or d.n % 128 == 0) and (tile_n != 256 or d.n % 256 == 0)) else 0
        ldt = dtype
        if dtype in (5, 7) and (d.x_stride_w % 32 or d.cin % 32):
            ldt = 4  # stem: f32 image in, sp32 out
        m = d.batch * d.out_h * d.out_w
        k = d.kh * d.kw * d.cin
        x = torch.randn(max(L["in_elems"], d.x_stride_b * d.batch) + 64, device=engine.device).to(tin)
        g = L.get("groups", 1)
        w = (torch.randn(g * d.n * k, device=engine.device) / k ** 0.5).to(tin)
        if dtype == 7:  # "x3w": the weights-direct form wherever the shape allows it, else the staged form (as the library does)
            if ldt == 7 and g == 1 and d.n % 256 == 0 and (k // 32) % 2 == 0:
                w = engine.weight_frags(w.reshape(d.n, k))
            else:
                ldt = 5 if ldt == 7 else ldt
                w = engine.split_weight_rows(w.reshape(g * d.n, k))
        elif dtype >= 3:
            w = engine.split_weight_rows(w.reshape(g * d.n, k))
        if dtype in (4, 5, 7) and d.y_ld % 32:
            continue
        ylen = m * max(d.y_ld, d.n) + 64
        y = torch.empty(ylen, device=engine.device, dtype=tin)
        res = torch.randn(ylen, device=engine.device).to(tin) if L["res"] else None
        sc, bi = torch.ones(g * d.n, device=engine.device), torch.zeros(g * d.n, device=engine.device)
        for _ in range(2):
            engine.conv_gemm(d, ldt, x, w, sc, bi, res, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            engine.conv_gemm(d, ldt, x, w, sc, bi, res, y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        fl = 2.0 * m * d.n * k * g
        by = (L["in_elems"] + L["out_elems"] * (2 if L["res"] else 1) + d.n * k) * es
        print(f"{L['name']:38s} x{L['count']:<2d} M={m:<7d} N={d.n:<5d} K={k:<5d} {ms*1e3:9.1f} us {fl/ms/1e9:8.1f} TF/s "
              f"{by/ms/1e6:8.1f} GB/s  blocks={((m+127)//128)*(d.n//(128 if d.n%128==0 else 64))}")
        tot_ms += ms * L["count"]
        tot_fl += fl * L["count"]
    print(f"=== {title}: {tot_ms:.2f} ms, {tot_fl/tot_ms/1e9:.1f} TF/s on executed FLOPs")
    return tot_ms


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="both")
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--chunks", type=int, default=128)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--tile-n", type=int, default=0, help="force the block tile width (64 / 128) on every layer: a tuning sweep")
    ap.add_argument("--no-audio", action="store_true")
    a = ap.parse_args()
    eng = Engine(0)
    for name, dt in (("f32", 0), ("bf16", 1), ("x3", 3), ("x3s", 5), ("x3w", 7)):
        if a.dtype in ("both", name):
            run(eng, static_layers(a.frames), dt, a.iters, f"static CNN, {a.frames} frames, {name}, tile_n={a.tile_n}", a.tile_n)
            if not a.no_audio:
                run(eng, audio_layers(a.chunks, 32000), dt, a.iters, f"audio model, {a.chunks} x 2 s, {name}, tile_n={a.tile_n}", a.tile_n)
