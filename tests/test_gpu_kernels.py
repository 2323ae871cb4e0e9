"""Kernel-level parity of the contraction kernel (avcer_conv_gemm) against a float64 torch reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from avcer_amd._lib import ConvDesc
from avcer_amd.sp32 import from_sp32, split_weight_mul, to_sp32

pytestmark = pytest.mark.gpu


def _desc(**kw):
    d = ConvDesc()
    base = dict(batch=1, in_h=1, in_w=1, out_h=1, out_w=1, cin=32, kh=1, kw=1, stride_h=1, stride_w=1, pad_h=0, pad_w=0,
                dil_h=1, dil_w=1, x_stride_b=0, x_stride_h=0, x_stride_w=0, x_coff=0, n=64, y_ld=64, y_coff=0, r_ld=64,
                r_coff=0, act=0, res_after_act=0, groups=0, x2_cin=0, x2_coff=0, x2_stride=0, x2_stride_b=0, x2_stride_h=0,
                x2_stride_w=0, tile_n=0)
    base.update(kw)
    for k, v in base.items():
        setattr(d, k, int(v))
    return d


def _act(v, act):
    return F.relu(v) if act == 1 else (F.gelu(v) if act in (2, 3) else v)


A_KIND = {0: "f32", 1: "bf16", 2: "bf16", 3: "f32", 4: "f32", 5: "sp32", 6: "sp32"}
O_KIND = {0: "f32", 1: "bf16", 2: "f32", 3: "f32", 4: "sp32", 5: "sp32", 6: "f32"}


def _enc(t, kind, dev):
    if kind == "f32":
        return t.to(dev, torch.float32).contiguous()
    if kind == "bf16":
        return t.to(dev, torch.bfloat16).contiguous()
    return to_sp32(t.float()).to(dev)


def _dec(t, kind):
    return from_sp32(t.cpu()) if kind == "sp32" else t.float().cpu()


def _run(engine, d, dtype, x, w, scale, bias, res, y):
    dev = engine.device
    ak, ok = A_KIND[dtype], O_KIND[dtype]
    xd = _enc(x, ak, dev)
    wd = w.to(dev, torch.bfloat16 if ak == "bf16" else torch.float32).contiguous()
    # x3 arithmetic: scaled fp16 hi/lo split of the f32 weights, rows permuted inside groups of 32 output channels
    w_arg = engine.split_weight_rows(wd.reshape(wd.shape[0], -1)) if dtype >= 3 else wd
    sd_ = None if scale is None else scale.to(dev, torch.float32)
    bd = None if bias is None else bias.to(dev, torch.float32)
    rd = None if res is None else _enc(res, ok, dev)
    yd = _enc(y, ok, dev)
    engine.conv_gemm(d, dtype, xd, w_arg, sd_, bd, rd, yd)
    torch.cuda.synchronize()
    return _dec(yd, ok), _dec(xd, ak).double(), wd.double().cpu(), (None if rd is None else _dec(rd, ok).double())


def _tol(dtype, ref):
    scale = float(ref.abs().max()) + 1e-6
    return {0: 2e-5, 1: 1.2e-2, 2: 2e-4, 3: 6e-5, 4: 8e-5, 5: 8e-5, 6: 6e-5}[dtype] * scale


@pytest.mark.parametrize("dtype", [0, 1, 2, 3, 4, 5, 6])
def test_identity_times_asymmetric_weight(engine, dtype):
    """A = I with an ASYMMETRIC W catches a row/col swap of the accumulator layout (exact small integers)."""
    k = 64 if dtype in (1, 2) else 32
    m, n = 128, 128
    x = torch.zeros(m, k)
    x[torch.arange(k), torch.arange(k)] = 1.0
    w = (torch.arange(n)[:, None] * 2 + torch.arange(k)[None, :] % 7).float()  # W[n,k], < 256+7: exact in bf16? no ->
    w = (torch.arange(n)[:, None] % 16 * 8 + torch.arange(k)[None, :] % 8).float()  # <= 127: exact in bf16
    d = _desc(batch=m, cin=k, x_stride_b=k, x_stride_h=k, x_stride_w=k, n=n, y_ld=n, r_ld=n)
    y, _, _, _ = _run(engine, d, dtype, x, w, None, None, None, torch.full((m, n), -1.0))
    ref = x @ w.t()
    assert torch.equal(y, ref), (y - ref).abs().max()


@pytest.mark.parametrize("dtype", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("m,k,n", [(300, 128, 192), (129, 64, 256), (1, 256, 64), (1000, 2048, 128)])
def test_linear(engine, dtype, m, k, n):
    g = torch.Generator().manual_seed(m + k + n)
    x, w = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5
    scale, bias = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g)
    d = _desc(batch=m, cin=k, x_stride_b=k, x_stride_h=k, x_stride_w=k, n=n, y_ld=n, r_ld=n, act=1)
    y, xd, wd, rd = _run(engine, d, dtype, x, w, scale, bias, res, torch.zeros(m, n))
    ref = F.relu((xd @ wd.t()) * scale.double() + bias.double() + rd)
    assert (y.double() - ref).abs().max() < _tol(dtype, ref)


CONVS = [
    # b, h, w, c, kh, kw, stride, pad, dil, n, act
    (2, 9, 9, 64, 3, 3, 1, 1, 1, 64, 1),
    (3, 7, 7, 64, 1, 1, 2, 0, 1, 128, 0),
    (2, 14, 14, 128, 3, 3, 1, 1, 1, 128, 1),
    (1, 55, 55, 64, 3, 3, 1, 1, 1, 64, 1),
    (2, 31, 1, 64, 5, 1, 3, 0, 2, 64, 0),    # Conv1d k5 s3 dil2 (audio head)
    (2, 40, 1, 64, 2, 1, 2, 0, 1, 64, 2),    # Conv1d k2 s2 + gelu (feature extractor)
]


@pytest.mark.parametrize("dtype", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("cfg", CONVS)
def test_conv(engine, dtype, cfg):
    b, h, w_, c, kh, kw, s, p, dil, n, act = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(b, h, w_, c, generator=g)
    w = torch.randn(n, kh, kw, c, generator=g) / (kh * kw * c) ** 0.5
    scale, bias = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g)
    sw, pw, dw = (1, 0, 1) if w_ == 1 else (s, p, dil)
    oh = (h + 2 * p - dil * (kh - 1) - 1) // s + 1
    ow = (w_ + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    res = torch.randn(b, oh, ow, n, generator=g)
    d = _desc(batch=b, in_h=h, in_w=w_, out_h=oh, out_w=ow, cin=c, kh=kh, kw=kw, stride_h=s, stride_w=sw, pad_h=p,
              pad_w=pw, dil_h=dil, dil_w=dw, x_stride_b=h * w_ * c, x_stride_h=w_ * c, x_stride_w=c, n=n, y_ld=n, r_ld=n,
              act=act)
    y, xd, wd, rd = _run(engine, d, dtype, x, w.reshape(n, -1), scale, bias, res, torch.zeros(b, oh, ow, n))
    ref = F.conv2d(xd.permute(0, 3, 1, 2), wd.reshape(n, kh, kw, c).permute(0, 3, 1, 2), None, (s, sw), (p, pw), (dil, dw))
    ref = ref.permute(0, 2, 3, 1) * scale.double() + bias.double() + rd
    ref = _act(ref, act)
    assert (y.double() - ref).abs().max() < _tol(dtype, ref)


@pytest.mark.parametrize("dtype", [0, 2, 3, 5, 6])
def test_grouped_slices_and_residual_after_act(engine, dtype):
    """pos-conv shape: one group of a [B,S,C] tensor, zero padding in time, output/residual written into a channel
    slice, y = gelu(conv + bias) + residual."""
    b, s, ctot, g0, cin, k = 2, 50, 256, 64, 64, 8
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(b, s, ctot, generator=gen)
    w = torch.randn(64, k, cin, generator=gen) / (k * cin) ** 0.5
    bias = torch.randn(64, generator=gen)
    res = torch.randn(b, s, ctot, generator=gen)
    y0 = torch.full((b, s, ctot), 7.0)
    d = _desc(batch=b, in_h=s, in_w=1, out_h=s, out_w=1, cin=cin, kh=k, kw=1, pad_h=k // 2, x_stride_b=s * ctot,
              x_stride_h=ctot, x_stride_w=ctot, x_coff=g0, n=64, y_ld=ctot, y_coff=128, r_ld=ctot, r_coff=128, act=2,
              res_after_act=1)
    y, xd, wd, rd = _run(engine, d, dtype, x, w.reshape(64, -1), None, bias, res, y0)
    xg = xd[:, :, g0:g0 + cin].permute(0, 2, 1)
    ref = F.conv1d(xg, wd.reshape(64, k, cin).permute(0, 2, 1), bias.double(), padding=k // 2)[:, :, :s]
    ref = F.gelu(ref).permute(0, 2, 1) + rd[:, :, 128:192]
    assert (y[:, :, 128:192].double() - ref).abs().max() < _tol(dtype, ref)
    assert torch.equal(y[:, :, :128], y0[:, :, :128]) and torch.equal(y[:, :, 192:], y0[:, :, 192:])


@pytest.mark.parametrize("dtype", [0, 1, 3, 4])
def test_stem_layout(engine, dtype):
    """7x7/2 stem as 8 tap rows x (8 pixels x 4 channels) over a zero-bordered 230x230x4 image."""
    from avcer_amd import packing, synth

    sd = synth.static_state_dict(42)
    wp = torch.from_numpy(packing.pack_static(sd)["stem.w"])
    gen = torch.Generator().manual_seed(9)
    img = torch.zeros(2, 230, 230, 4)
    img[:, 2:226, 2:226, :3] = torch.randn(2, 224, 224, 3, generator=gen) * 50
    d = _desc(batch=2, in_h=230, in_w=230, out_h=112, out_w=112, cin=32, kh=8, kw=1, stride_h=2, stride_w=2,
              x_stride_b=230 * 230 * 4, x_stride_h=230 * 4, x_stride_w=4, n=64, y_ld=64, r_ld=64)
    y, xd, wd, _ = _run(engine, d, dtype, img, wp, None, None, None, torch.zeros(2, 112, 112, 64))
    w = wd.reshape(64, 8, 8, 4)[:, :7, :7, :3].permute(0, 3, 1, 2)
    x = F.pad(xd[:, 2:226, 2:226, :3].permute(0, 3, 1, 2), [2, 3, 2, 3])
    ref = F.conv2d(x, w, None, 2).permute(0, 2, 3, 1)
    assert (y.double() - ref).abs().max() < _tol(dtype, ref)


def test_bad_shapes_are_rejected(engine):
    from avcer_amd._lib import AvcerError

    x = torch.zeros(4, 32, device=engine.device)
    d = _desc(batch=4, cin=32, x_stride_b=32, x_stride_h=32, x_stride_w=32, n=48, y_ld=48)
    with pytest.raises(AvcerError):
        engine.conv_gemm(d, 0, x, x, None, None, None, x)
    d = _desc(batch=4, cin=24, x_stride_b=24, x_stride_h=24, x_stride_w=24, n=64, y_ld=64)
    with pytest.raises(AvcerError):
        engine.conv_gemm(d, 0, x, x, None, None, None, x)


def test_split_fp16_is_f32_grade_and_far_more_accurate_than_bf16(engine):
    """The x3 contraction (fp16 hi/lo pairs, three MFMAs per product) must sit at the f32 MFMA's own error, three orders
    below plain bf16 (the bf16 split of rounds 1-3 read 30 x the f32 error here)."""
    g = torch.Generator().manual_seed(3)
    m, k, n = 512, 1024, 256
    x, w = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5
    d = _desc(batch=m, cin=k, x_stride_b=k, x_stride_h=k, x_stride_w=k, n=n, y_ld=n, r_ld=n)
    ref = x.double() @ w.double().t()
    errs = {}
    for dtype in (0, 2, 3):
        y, *_ = _run(engine, d, dtype, x, w, None, None, None, torch.zeros(m, n))
        errs[dtype] = float((y.double() - ref).abs().max())
    print("max|err| f32 / bf16 / split-fp16:", errs)
    assert errs[3] < 3 * errs[0] + 1e-7 and errs[3] < errs[2] / 1000


@pytest.mark.parametrize("dtype", [0, 2, 3, 6])
def test_grouped_launch_matches_grouped_conv1d(engine, dtype):
    """wav2vec2 pos-conv shape in one launch: groups of 64 channels, k taps, zero padding, gelu(conv+b) + residual."""
    b, s, groups, cin, k = 2, 37, 4, 64, 16
    ctot = groups * cin
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(b, s, ctot, generator=gen)
    w = torch.randn(ctot, k, cin, generator=gen) / (k * cin) ** 0.5   # [out, k, in/groups]
    bias = torch.randn(ctot, generator=gen)
    res = torch.randn(b, s, ctot, generator=gen)
    d = _desc(batch=b, in_h=s, in_w=1, out_h=s, out_w=1, cin=cin, kh=k, kw=1, pad_h=k // 2, x_stride_b=s * ctot,
              x_stride_h=ctot, x_stride_w=ctot, n=64, y_ld=ctot, r_ld=ctot, act=2, res_after_act=1, groups=groups)
    y, xd, wd, rd = _run(engine, d, dtype, x, w.reshape(ctot, -1), None, bias, res, torch.zeros(b, s, ctot))
    ref = F.conv1d(xd.permute(0, 2, 1), wd.reshape(ctot, k, cin).permute(0, 2, 1), bias.double(), padding=k // 2,
                   groups=groups)[:, :, :s]
    ref = F.gelu(ref).permute(0, 2, 1) + rd
    assert (y.double() - ref).abs().max() < _tol(dtype, ref)


@pytest.mark.parametrize("dtype", [0, 1, 3, 5])
def test_dual_source_fused_1x1(engine, dtype):
    """ResNet block 0: relu(conv3(T2) + downsample(X, stride 2)) as ONE contraction over K = [T2 | X strided]."""
    b, h, p_, cin, n, st = 2, 14, 64, 128, 256, 2
    oh = (h - 1) // st + 1
    g = torch.Generator().manual_seed(21)
    x = torch.randn(b, h, h, cin, generator=g)            # block input (downsample source)
    t2 = torch.randn(b, oh, oh, p_, generator=g)          # conv2 output (conv3 source)
    w3 = torch.randn(n, p_, generator=g) / p_ ** 0.5
    wd = torch.randn(n, cin, generator=g) / cin ** 0.5
    bias = torch.randn(n, generator=g)
    w = torch.cat([w3, wd], dim=1)
    d = _desc(batch=b, in_h=oh, in_w=oh, out_h=oh, out_w=oh, cin=p_, x_stride_b=oh * oh * p_, x_stride_h=oh * p_, x_stride_w=p_,
              n=n, y_ld=n, r_ld=n, act=1, x2_cin=cin, x2_stride=st, x2_stride_b=h * h * cin, x2_stride_h=h * cin, x2_stride_w=cin)
    dev = engine.device
    ak, ok = A_KIND[dtype], O_KIND[dtype]
    xd, td = _enc(x, ak, dev), _enc(t2, ak, dev)
    wdv = w.to(dev, torch.bfloat16 if ak == "bf16" else torch.float32).contiguous()
    w_arg = engine.split_weight_rows(wdv) if dtype >= 3 else wdv
    yd = _enc(torch.zeros(b, oh, oh, n), ok, dev)
    engine.conv_gemm_dual(d, dtype, td, xd, w_arg, None, bias.to(dev), None, yd)
    torch.cuda.synchronize()
    xs = _dec(xd, ak).double()[:, ::st, ::st, :]
    ref = F.relu(_dec(td, ak).double() @ wdv.double().cpu()[:, :p_].t() + xs @ wdv.double().cpu()[:, p_:].t() + bias.double())
    got = _dec(yd, ok).double()
    assert (got - ref).abs().max() < _tol(dtype, ref)


@pytest.mark.parametrize("dtype", [0, 5, 7])
def test_gelu_with_the_short_erf(engine, dtype):
    """act = 3: GELU with the Abramowitz-Stegun 7.1.26 erf (gemm_dev.h gelu_fast).  An identity contraction turns the epilogue
    into a function evaluator: 64 k arguments across [-8, 8] against float64 GELU (audio_8_cl.py / wav2vec2's nn.GELU, erf
    form).  Bound: 5e-7 absolute (the exact form's own f32 rounding reaches 4.5e-7), plus the sp32 encoding step where
    the output is sp32."""
    n = 256
    m = 256
    g = torch.Generator().manual_seed(5)
    xs = torch.cat([torch.linspace(-8, 8, m * n - 4096), torch.randn(4096, generator=g) * 0.01]).reshape(m, n)
    w = torch.eye(n)
    d = _desc(batch=m, cin=n, x_stride_b=n, x_stride_h=n, x_stride_w=n, n=n, y_ld=n, r_ld=n, act=3)
    dev = engine.device
    ak, ok = ("f32", "f32") if dtype == 0 else ("sp32", "sp32")
    xd = _enc(xs, ak, dev)
    wd = w.to(dev)
    w_arg = wd if dtype == 0 else (engine.weight_frags(wd) if dtype == 7 else engine.split_weight_rows(wd))
    yd = _enc(torch.zeros(m, n), ok, dev)
    engine.conv_gemm(d, dtype, xd, w_arg, None, None, None, yd)
    torch.cuda.synchronize()
    arg = _dec(xd, ak).double()                       # what the kernel saw (sp32 rounds the arguments to 2^-17)
    ref = 0.5 * arg * (1.0 + torch.erf(arg / 2 ** 0.5))
    got = _dec(yd, ok).double()
    err = (got - ref).abs()
    tol = 5e-7 + (0 if dtype == 0 else 2.0 ** -16 * ref.abs().max().item())
    print(f"gelu act=3 dtype {dtype}: max abs err {err.max().item():.2e}")
    assert err.max().item() < tol


# ----------------------------------------------------------------------------- fused bottleneck chain (csrc/fused.hip)
@pytest.mark.parametrize("planes,nb,hw,nxt", [(64, 3, 55, True), (64, 2, 9, False), (128, 5, 28, True), (128, 1, 7, False)])
def test_bneck_chain_vs_float64(engine, planes, nb, hw, nxt):
    """conv3x3+ReLU -> conv1x1 + residual + ReLU -> next conv1x1 + ReLU in one launch against float64 torch convolutions
    (video.py:43-60 with folded BatchNorm).  M = nb*hw*hw is never a multiple of the 128-position tile here."""
    g = torch.Generator().manual_seed(planes + hw)
    p4 = 4 * planes
    t1 = torch.rand(nb, hw, hw, planes, generator=g) * 2          # post-ReLU activations
    x = torch.rand(nb, hw, hw, p4, generator=g) * 2
    w2 = torch.randn(planes, 3, 3, planes, generator=g) / (3 * planes ** 0.5)   # [O][kh][kw][I]
    w3 = torch.randn(p4, planes, generator=g) / planes ** 0.5
    w1 = torch.randn(planes, p4, generator=g) / p4 ** 0.5
    b2, b3, b1 = (torch.randn(n, generator=g) * 0.3 for n in (planes, p4, planes))
    # float64 reference (NCHW convolutions)
    t2 = F.relu(F.conv2d(t1.permute(0, 3, 1, 2).double(), w2.permute(0, 3, 1, 2).double(), b2.double(), padding=1))
    out = F.relu(F.conv2d(t2, w3.double()[:, :, None, None], b3.double()) + x.permute(0, 3, 1, 2).double())
    t1n = F.relu(F.conv2d(out, w1.double()[:, :, None, None], b1.double()))
    dev = engine.device

    def wsplit(w):
        return engine.split_weight_rows(w.reshape(w.shape[0], -1))

    d_out = torch.full((nb, hw, hw, 2 * p4), 0x7fc0, dtype=torch.int16, device=dev)   # NaN-filled sp32
    d_t1n = torch.full((nb, hw, hw, 2 * planes), 0x7fc0, dtype=torch.int16, device=dev) if nxt else None
    engine.bneck_chain(planes, nb, hw, hw, to_sp32(t1).to(dev), to_sp32(x).to(dev), d_out, d_t1n, wsplit(w2), b2.to(dev),
                       wsplit(w3), b3.to(dev), wsplit(w1) if nxt else None, b1.to(dev) if nxt else None)
    torch.cuda.synchronize()
    got = from_sp32(d_out.cpu()).permute(0, 3, 1, 2).double()
    err = (got - out).abs().max().item()
    print(f"bneck planes={planes} {nb}x{hw}x{hw}: max|out err| {err:.2e} (max|out| {out.abs().max().item():.1f})")
    assert err < 2e-5 * max(1.0, out.abs().max().item())
    if nxt:
        got1 = from_sp32(d_t1n.cpu()).permute(0, 3, 1, 2).double()
        err1 = (got1 - t1n).abs().max().item()
        print(f"   max|t1n err| {err1:.2e} (max {t1n.abs().max().item():.1f})")
        assert err1 < 2e-5 * max(1.0, t1n.abs().max().item())


@pytest.mark.parametrize("planes,nb,hw", [(64, 3, 55), (64, 2, 9), (128, 5, 28), (128, 3, 7)])
def test_bneck_chain_last_block_even_positions(engine, planes, nb, hw):
    """out_step = 2, the last block of a stage: only the positions (2 oy, 2 ox) that the next stage's stride-2 1x1
    convolutions read (video.py:12-19,140-149) are evaluated.  Every value must be BIT-IDENTICAL to what the full-resolution
    launch puts at that position (same taps, same K order) -- except where the full launch takes the resident-patch form of
    the conv2 phase (planes 128 on images of >= 128 positions), which walks K as (channel chunk, tap) instead of (tap,
    channel chunk): there the two agree to one step of the sp32 encoding.  Both are also checked against float64 convolutions."""
    g = torch.Generator().manual_seed(planes + hw + 1)
    p4 = 4 * planes
    t1 = torch.rand(nb, hw, hw, planes, generator=g) * 2
    x = torch.rand(nb, hw, hw, p4, generator=g) * 2
    w2 = torch.randn(planes, 9 * planes, generator=g) / (3 * planes ** 0.5)
    w3 = torch.randn(p4, planes, generator=g) / planes ** 0.5
    b2, b3 = (torch.randn(n, generator=g) * 0.3 for n in (planes, p4))
    dev = engine.device
    oh = (hw + 1) // 2
    args = (to_sp32(t1).to(dev), to_sp32(x).to(dev))
    w = (engine.split_weight_rows(w2), b2.to(dev), engine.split_weight_rows(w3), b3.to(dev))
    full = torch.full((nb, hw, hw, 2 * p4), 0x7fc0, dtype=torch.int16, device=dev)
    even = torch.full((nb, oh, oh, 2 * p4), 0x7fc0, dtype=torch.int16, device=dev)
    engine.bneck_chain(planes, nb, hw, hw, *args, full, None, *w)
    engine.bneck_chain(planes, nb, hw, hw, *args, even, None, *w, out_step=2)
    torch.cuda.synchronize()
    got, ref_full = from_sp32(even.cpu()), from_sp32(full.cpu()[:, ::2, ::2])
    if planes == 128 and hw * hw >= 128:
        assert (got - ref_full).abs().max() < 2e-5 * ref_full.abs().max()  # one step of the sp32 encoding (2^-16) at most
    else:
        assert torch.equal(even.cpu(), full.cpu()[:, ::2, ::2])
    t2 = F.relu(F.conv2d(t1.permute(0, 3, 1, 2).double(), w2.reshape(planes, 3, 3, planes).permute(0, 3, 1, 2).double(), b2.double(),
                         padding=1))
    out = F.relu(F.conv2d(t2, w3.double()[:, :, None, None], b3.double()) + x.permute(0, 3, 1, 2).double())[:, :, ::2, ::2]
    err = (got.permute(0, 3, 1, 2).double() - out).abs().max().item()
    print(f"bneck last block planes={planes} {nb}x{hw}x{hw} -> {oh}x{oh}: max|out err| {err:.2e} (max|out| {out.abs().max().item():.1f})")
    assert err < 2e-5 * max(1.0, out.abs().max().item())


def test_bneck_chain_strided_form_refuses_a_next_conv1(engine):
    from avcer_amd._lib import AvcerError

    dev = engine.device
    z = torch.zeros(1, 8, 8, 128, dtype=torch.int16, device=dev)
    o = torch.zeros(1, 4, 4, 512, dtype=torch.int16, device=dev)
    w = torch.zeros(64 * 576 * 2, dtype=torch.int16, device=dev)
    b = torch.zeros(256, device=dev)
    with pytest.raises(AvcerError, match="last block of a stage"):
        engine.bneck_chain(64, 1, 8, 8, z, o, o, z, w, b, w, b, w, b, out_step=2)


@pytest.mark.parametrize("dtype", [0, 1, 5, 7])
def test_sub_sampled_residual(engine, dtype):
    """avcer_conv_desc.r_sub = 2: conv3 of a stage's last block evaluated on the compact even grid, the residual row taken
    from position (2 oy, 2 ox) of the full-resolution block input (video.py:49-58 at the positions video.py:12-19 reads)."""
    b, h, k, n = 3, 13, 64, 256
    oh = (h + 1) // 2
    g = torch.Generator().manual_seed(31 + dtype)
    t2 = torch.randn(b, oh, oh, k, generator=g)
    x = torch.randn(b, h, h, n, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    scale, bias = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g)
    d = _desc(batch=b, in_h=oh, in_w=oh, out_h=oh, out_w=oh, cin=k, x_stride_b=oh * oh * k, x_stride_h=oh * k, x_stride_w=k,
              n=n, y_ld=n, r_ld=n, act=1, r_sub=2, r_h=h, r_w=h)
    dev = engine.device
    base = {7: 5}.get(dtype, dtype)
    ak, ok = A_KIND[base], O_KIND[base]
    td, rd = _enc(t2, ak, dev), _enc(x, ok, dev)
    wdv = w.to(dev, torch.bfloat16 if ak == "bf16" else torch.float32).contiguous()
    w_arg = wdv if dtype < 3 else (engine.weight_frags(wdv) if dtype == 7 else engine.split_weight_rows(wdv))
    yd = _enc(torch.zeros(b, oh, oh, n), ok, dev)
    engine.conv_gemm(d, dtype, td, w_arg, scale.to(dev), bias.to(dev), rd, yd)
    torch.cuda.synchronize()
    ref = F.relu((_dec(td, ak).double() @ wdv.double().cpu().t()) * scale.double() + bias.double() + _dec(rd, ok).double()[:, ::2, ::2])
    got = _dec(yd, ok).double()
    assert (got - ref).abs().max() < _tol(base, ref)
    # too small a residual grid is refused
    from avcer_amd._lib import AvcerError
    d.r_h = 2 * oh - 2
    with pytest.raises(AvcerError, match="residual grid"):
        engine.conv_gemm(d, dtype, td, w_arg, scale.to(dev), bias.to(dev), rd, yd)


def test_split_weight_rows_layout(engine):
    """avcer_split_weight_rows: hi/lo per 32-element K group + the row order of packing.permute_rows_for_mfma."""
    from avcer_amd.packing import permute_rows_for_mfma

    w = torch.randn(96, 64, generator=torch.Generator().manual_seed(1)) * 0.03
    buf = engine.split_weight_rows(w).cpu()
    mul = split_weight_mul(buf, 96, 64)
    # one power of two per matrix: the largest magnitude lands in [2^14, 2^15)
    assert mul == 2.0 ** round(np.log2(mul)) and 2 ** 14 <= w.abs().max().item() / mul < 2 ** 15
    got = from_sp32(buf[:96 * 128].reshape(96, 128))
    ref = torch.from_numpy(permute_rows_for_mfma(w.numpy())) / mul
    assert (got - ref).abs().max() <= 2.0 ** -22 * ref.abs().max()             # 11 + 11 significand bits
    assert torch.equal(from_sp32(to_sp32(ref)), got)
    # the fragment-order copy carries the same trailer
    assert split_weight_mul(engine.weight_frags(w).cpu(), 96, 64) == mul


def test_bneck_chain_first_block_with_downsample(engine):
    """The first block of stage 1 (video.py:43-60 with i_downsample, stride 1): conv3 and the downsample convolution as one
    contraction over K = [T2 | x], no residual, then the next block's conv1 -- against float64 convolutions."""
    g = torch.Generator().manual_seed(7)
    planes, nb, hw = 64, 3, 55
    p4 = 4 * planes
    t1 = torch.rand(nb, hw, hw, planes, generator=g) * 2
    x = torch.rand(nb, hw, hw, 64, generator=g) * 2
    w2 = torch.randn(planes, 3, 3, planes, generator=g) / (3 * planes ** 0.5)
    w3 = torch.randn(p4, planes, generator=g) / planes ** 0.5
    wd = torch.randn(p4, 64, generator=g) / 8.0
    w1 = torch.randn(planes, p4, generator=g) / p4 ** 0.5
    b2, b3, b1 = (torch.randn(n, generator=g) * 0.3 for n in (planes, p4, planes))
    t2 = F.relu(F.conv2d(t1.permute(0, 3, 1, 2).double(), w2.permute(0, 3, 1, 2).double(), b2.double(), padding=1))
    out = F.relu(F.conv2d(t2, w3.double()[:, :, None, None], b3.double()) +
                 F.conv2d(x.permute(0, 3, 1, 2).double(), wd.double()[:, :, None, None]))
    t1n = F.relu(F.conv2d(out, w1.double()[:, :, None, None], b1.double()))
    dev = engine.device
    d_out = torch.full((nb, hw, hw, 2 * p4), 0x7fc0, dtype=torch.int16, device=dev)
    d_t1n = torch.full((nb, hw, hw, 2 * planes), 0x7fc0, dtype=torch.int16, device=dev)
    engine.bneck_chain(planes, nb, hw, hw, to_sp32(t1).to(dev), to_sp32(x).to(dev), d_out, d_t1n,
                       engine.split_weight_rows(w2.reshape(planes, -1)), b2.to(dev),
                       engine.split_weight_rows(torch.cat([w3, wd], dim=1)), b3.to(dev), engine.split_weight_rows(w1), b1.to(dev),
                       ds_cin=64)
    torch.cuda.synchronize()
    e0 = (from_sp32(d_out.cpu()).permute(0, 3, 1, 2).double() - out).abs().max().item()
    e1 = (from_sp32(d_t1n.cpu()).permute(0, 3, 1, 2).double() - t1n).abs().max().item()
    print(f"bneck first block: max|out err| {e0:.2e} (max {out.abs().max().item():.1f}), max|t1n err| {e1:.2e}")
    assert e0 < 2e-5 * max(1.0, out.abs().max().item()) and e1 < 2e-5 * max(1.0, t1n.abs().max().item())


@pytest.mark.parametrize("ds", [False, True])
@pytest.mark.parametrize("nb", [1, 3])
def test_bneck_chain_spatial_tile_form(engine, nb, ds):
    """The spatial-tile form of the planes-64 chain (bneck_kernel<..., T11>: 11 x 11 tiles of the 55 x 55 image, resident
    13 x 13 halo patch, conv2 weight fragments loaded straight into registers -- selected by passing `w2_frags`): every output
    element must be BIT-IDENTICAL to the gather form's (same products in the same order), image borders, tile seams and the
    seven idle rows of every tile included; and both are held to float64 convolutions.  With and without the downsample
    operand of a stage's first block (video.py:43-60)."""
    g = torch.Generator().manual_seed(100 + nb + int(ds))
    planes, hw = 64, 55
    p4, xc = 4 * planes, (64 if ds else 4 * planes)
    t1 = torch.rand(nb, hw, hw, planes, generator=g) * 2
    x = torch.rand(nb, hw, hw, xc, generator=g) * 2
    w2 = torch.randn(planes, 3, 3, planes, generator=g) / (3 * planes ** 0.5)
    w3 = torch.randn(p4, planes, generator=g) / planes ** 0.5
    wd = torch.randn(p4, 64, generator=g) / 8.0
    w1 = torch.randn(planes, p4, generator=g) / p4 ** 0.5
    b2, b3, b1 = (torch.randn(n, generator=g) * 0.3 for n in (planes, p4, planes))
    t2 = F.relu(F.conv2d(t1.permute(0, 3, 1, 2).double(), w2.permute(0, 3, 1, 2).double(), b2.double(), padding=1))
    o = F.conv2d(t2, w3.double()[:, :, None, None], b3.double())
    o = o + (F.conv2d(x.permute(0, 3, 1, 2).double(), wd.double()[:, :, None, None]) if ds else x.permute(0, 3, 1, 2).double())
    out = F.relu(o)
    t1n = F.relu(F.conv2d(out, w1.double()[:, :, None, None], b1.double()))
    dev = engine.device
    w2m = w2.reshape(planes, -1).to(dev)
    w3m = torch.cat([w3, wd], dim=1) if ds else w3
    args = (to_sp32(t1).to(dev), to_sp32(x).to(dev))
    wts = (engine.split_weight_rows(w2m), b2.to(dev), engine.split_weight_rows(w3m), b3.to(dev), engine.split_weight_rows(w1), b1.to(dev))
    res = []
    for frags in (None, engine.weight_frags(w2m)):
        d_out = torch.full((nb, hw, hw, 2 * p4), 0x7fc0, dtype=torch.int16, device=dev)
        d_t1n = torch.full((nb, hw, hw, 2 * planes), 0x7fc0, dtype=torch.int16, device=dev)
        engine.bneck_chain(planes, nb, hw, hw, *args, d_out, d_t1n, *wts, ds_cin=64 if ds else 0, w2_frags=frags)
        torch.cuda.synchronize()
        res.append((d_out.cpu(), d_t1n.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    e0 = (from_sp32(res[1][0]).permute(0, 3, 1, 2).double() - out).abs().max().item()
    e1 = (from_sp32(res[1][1]).permute(0, 3, 1, 2).double() - t1n).abs().max().item()
    print(f"bneck spatial-tile form nb={nb} ds={ds}: max|out err| {e0:.2e}, max|t1n err| {e1:.2e}")
    assert e0 < 2e-5 * max(1.0, out.abs().max().item()) and e1 < 2e-5 * max(1.0, t1n.abs().max().item())


def test_stem_pool_vs_float64(engine):
    """conv 7x7/2 with TF-"same" padding (2 before, 3 after) + BN + ReLU + max-pool 3x3/2 (video.py:63-90,98-103,116-117) in
    one launch, from the planar fp16 hi/lo image, against float64 torch ops."""
    g = torch.Generator().manual_seed(11)
    n = 3
    img = torch.randn(n, 3, 224, 224, generator=g) * 60.0                 # preprocessed BGR - mean values
    w = torch.randn(64, 3, 7, 7, generator=g) / (147 ** 0.5)
    scale, bias = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    x = F.pad(img.double(), (2, 3, 2, 3))
    ref = F.max_pool2d(F.relu(F.conv2d(x, w.double(), stride=2) * scale.double()[None, :, None, None] + bias.double()[None, :, None, None]),
                       3, 2)                                               # [n, 64, 55, 55]
    # zero-bordered NHWC4 image (border = the padding, 4th channel 0), split into fp16 hi / lo planes
    pad = torch.zeros(n, 230, 230, 4)
    pad[:, 2:226, 2:226, :3] = img.permute(0, 2, 3, 1)
    hi = pad.to(torch.float16)
    lo = (pad - hi.float()).to(torch.float16)
    planes = torch.stack([hi.view(torch.int16), lo.view(torch.int16)])
    w7 = torch.zeros(64, 7, 8, 4)
    w7[:, :, :7, :3] = w.permute(0, 2, 3, 1)                               # [O][kh][kw][I] with kw, I zero-padded to 8 x 4
    dev = engine.device
    y = engine.stem_pool(planes.to(dev), engine.split_weight_rows(w7.reshape(64, 224)), scale.to(dev), bias.to(dev), n)
    torch.cuda.synchronize()
    got = from_sp32(y.cpu()).permute(0, 3, 1, 2).double()
    err = (got - ref).abs().max().item()
    print(f"stem_pool: max|err| {err:.2e} (max|ref| {ref.abs().max().item():.1f})")
    assert err < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("hw", [(224, 224), (180, 250)])
def test_stem_pool_from_u8_frames_vs_float64(engine, hw):
    """The stem fed with the u8 frames: NEAREST resize to 224 x 224 (PIL rule), BGR flip and mean subtraction
    (data/utils.py:19-39) happen inside, the pixel values enter the MFMA as exact bf16 integers and the means live in the 9
    border-class shifts of packing.stem_border_shifts.  Against float64: preprocessing as the reference does it, Conv2dSame's
    padding of the NORMALISED image (video.py:68-80), BN, ReLU, max-pool."""
    from avcer_amd.packing import PIXEL_MEANS, stem_border_shifts

    h, w_ = hw
    g = torch.Generator().manual_seed(13 + h)
    n = 3
    frames = torch.randint(0, 256, (n, h, w_, 3), generator=g, dtype=torch.uint8)
    frames[0, : h // 3] = 255                                               # saturated and black regions reach the borders
    frames[1, :, : w_ // 4] = 0
    w = torch.randn(64, 3, 7, 7, generator=g) / (147 ** 0.5)
    scale, bias = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    # reference preprocessing: PIL NEAREST = src index floor((dst + 0.5) * in / out), RGB -> BGR, minus the channel means
    sy = torch.clamp(((torch.arange(224).double() + 0.5) * (h / 224.0)).floor().long(), max=h - 1)
    sx = torch.clamp(((torch.arange(224).double() + 0.5) * (w_ / 224.0)).floor().long(), max=w_ - 1)
    img = frames[:, sy][:, :, sx].double().flip(-1) - torch.tensor(PIXEL_MEANS, dtype=torch.float64)   # [n,224,224,3] BGR - mean
    x = F.pad(img.permute(0, 3, 1, 2), (2, 3, 2, 3))
    ref = F.max_pool2d(F.relu(F.conv2d(x, w.double(), stride=2) * scale.double()[None, :, None, None] + bias.double()[None, :, None, None]),
                       3, 2)
    w7 = torch.zeros(64, 7, 8, 4)
    w7[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    dev = engine.device
    shifts = torch.from_numpy(stem_border_shifts(w.numpy(), scale.numpy(), bias.numpy()))
    y = engine.stem_pool_u8(frames.to(dev), engine.split_weight_rows(w7.reshape(64, 224)), scale.to(dev), shifts.to(dev))
    torch.cuda.synchronize()
    got = from_sp32(y.cpu()).permute(0, 3, 1, 2).double()
    err = (got - ref).abs().max().item()
    print(f"stem_pool_u8 {h}x{w_}: max|err| {err:.2e} (max|ref| {ref.abs().max().item():.1f})")
    assert err < 2e-5 * max(1.0, ref.abs().max().item())


# ---- range contract of the fp16 split (csrc/split_dev.h): activations unscaled, |x| < 65504; overflow -> NaN, never garbage
@pytest.mark.parametrize("dtype", [5, 7])
def test_x3_keeps_f32_grade_accuracy_with_activations_near_1e4(engine, dtype):
    """sp32 activations of magnitude 1e4 (fp16 tops out at 65504) through a contraction whose outputs are of that order
    too: the error stays at the f32 MFMA's level relative to the result."""
    g = torch.Generator().manual_seed(5)
    m, k, n = 384, 512, 256
    x = torch.rand(m, k, generator=g) * 3.0e4                      # 0 .. 3e4
    w = torch.randn(n, k, generator=g) / k                         # outputs: rms ~ 7e2, max ~ 4e3
    bias = torch.randn(n, generator=g) * 1.0e4
    d = _desc(batch=m, cin=k, x_stride_b=k, x_stride_h=k, x_stride_w=k, n=n, y_ld=n, r_ld=n, act=1)
    dev = engine.device
    xd = to_sp32(x).to(dev)
    yd = torch.zeros(m, 2 * n, dtype=torch.int16, device=dev)
    w_arg = engine.weight_frags(w.to(dev)) if dtype == 7 else engine.split_weight_rows(w.to(dev))
    engine.conv_gemm(d, dtype, xd, w_arg, None, bias.to(dev), None, yd)
    torch.cuda.synchronize()
    ref = F.relu(from_sp32(xd.cpu()).double() @ w.double().t() + bias.double())
    got = from_sp32(yd.cpu()).double()
    assert ref.max() > 2.0e4 and torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    print(f"dtype {dtype}: activations up to 3e4, outputs up to {ref.max().item():.0f}: max|err| {err:.2e}")
    assert err < 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("mag", [1e-2, 1e-3])
@pytest.mark.parametrize("dtype", [5, 7, 9])
def test_x3_small_magnitude_activations_keep_the_absolute_bound(engine, dtype, mag):
    """The OTHER end of the range contract (include/avcer_hip.h): a whole activation tensor of magnitude 1e-2 / 1e-3.  Below
    2^-3 the lo half of a pair is an fp16 subnormal (quantum 2^-24), so an element carries an ABSOLUTE error of at most 2^-25
    instead of the 2^-22 relative one -- 3e-6 relative at 1e-2, 3e-5 at 1e-3.  Asserted: (a) the stored pair is within 2^-25
    of the f32 value (host split and the device's own epilogue split alike: dtype 4 writes what dtype 5 reads), (b) a
    contraction over such a tensor is within sum_k |w_k| * 2^-25 of the exact result plus the f32 accumulation's own error --
    the subnormal halves are neither flushed by the conversion nor by the MFMA."""
    g = torch.Generator().manual_seed(int(1 / mag) + dtype)
    m, k, n = 256, 512, 256
    x = (torch.rand(m, k, generator=g) + 0.5) * mag * torch.where(torch.rand(m, k, generator=g) < 0.5, -1.0, 1.0)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    bound = 2.0 ** -25
    assert (from_sp32(to_sp32(x)).double() - x.double()).abs().max().item() <= bound       # (a) host split
    dev = engine.device
    # (a) device split: an identity contraction in dtype 4 (f32 in -> sp32 out) stores x through the kernels' own sp_value path
    eye = torch.eye(k)
    d_id = _desc(batch=m, cin=k, x_stride_b=k, x_stride_h=k, x_stride_w=k, n=k, y_ld=k, r_ld=k)
    xs = torch.zeros(m, 2 * k, dtype=torch.int16, device=dev)
    engine.conv_gemm(d_id, 4, x.to(dev), engine.split_weight_rows(eye.to(dev)), None, None, None, xs)
    torch.cuda.synchronize()
    dev_split_err = (from_sp32(xs.cpu()).double() - x.double()).abs().max().item()
    # the identity contraction itself re-adds hi + lo of x on the f32 accumulator: exact for one non-zero product per output
    assert dev_split_err <= 2 * bound, dev_split_err
    # (b) the contraction over the small tensor
    d = _desc(batch=m, cin=k, x_stride_b=k, x_stride_h=k, x_stride_w=k, n=n, y_ld=n, r_ld=n, tile_m=0)
    yd = torch.zeros(m, 2 * n, dtype=torch.int16, device=dev)
    w_arg = engine.weight_frags(w.to(dev)) if dtype in (7, 9) else engine.split_weight_rows(w.to(dev))
    engine.conv_gemm(d, dtype, to_sp32(x).to(dev), w_arg, None, None, None, yd)
    torch.cuda.synchronize()
    ref = x.double() @ w.double().t()
    got = from_sp32(yd.cpu()).double()
    err = (got - ref).abs().max().item()
    budget = w.abs().sum(1).max().item() * bound + 2e-6 * ref.abs().max().item() + bound
    rel = ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"dtype {dtype}, |x| ~ {mag:g}: pair error {dev_split_err:.2e} (2^-25 = {bound:.2e}); contraction max|err| {err:.2e} "
          f"(budget {budget:.2e}), rms relative {rel:.2e}")
    assert err <= budget
    # and in practice far inside it: the element errors are independent (uniform in +-2^-25: 1.7e-8 rms), so the result carries
    # the elements' own relative rms error -- 1.7e-6 at |x| ~ 1e-2, 1.7e-5 at 1e-3 (measured 2.4e-6 / 2.4e-5), against 7e-8 for
    # activations inside [2^-3, 65504): small-magnitude TENSORS are where the x3 mode is less than f32-grade
    assert rel < (4e-6 if mag >= 1e-2 else 4e-5)


def test_x3_overflow_is_nan_not_a_wrong_number(engine):
    """An sp32 OUTPUT beyond fp16's range: hi = +-inf, lo = x - inf = -+inf, hi + lo = NaN for every consumer."""
    m, k, n = 128, 64, 64
    x = torch.full((m, k), 2.0e3)
    w = torch.ones(n, k)                                            # outputs 1.28e5 > 65504
    w[1::2] *= 0.25                                                 # odd channels 3.2e4: representable
    d = _desc(batch=m, cin=k, x_stride_b=k, x_stride_h=k, x_stride_w=k, n=n, y_ld=n, r_ld=n)
    dev = engine.device
    yd = torch.zeros(m, 2 * n, dtype=torch.int16, device=dev)
    engine.conv_gemm(d, 5, to_sp32(x).to(dev), engine.split_weight_rows(w.to(dev)), None, None, None, yd)
    torch.cuda.synchronize()
    got = from_sp32(yd.cpu())
    assert torch.isnan(got[:, 0::2]).all() and (got[:, 1::2] == 3.2e4).all()
    # ... and as the INPUT of the next contraction the NaN spreads to every output that reads it
    y2 = torch.zeros(m, n, device=dev)
    engine.conv_gemm(d, 6, yd, engine.split_weight_rows(torch.ones(n, n, device=dev)), None, None, None, y2)
    torch.cuda.synchronize()
    assert torch.isnan(y2).all()


# ---- attention kernel on its own (avcer_attention)
@pytest.mark.parametrize("s,heads,d", [(99, 16, 64), (199, 4, 64), (99, 8, 32), (37, 2, 64)])
def test_attention_f32_and_x3_against_float64(engine, s, heads, d):
    """softmax(Q K^T * scale) V per head (attention_layers.py:80-144; transformers Wav2Vec2Attention): the f32 VALU kernel
    and the x3 MFMA kernel (fp16 hi/lo pairs, sp32 output) against float64.  Both must be f32-grade -- the x3 form read
    6.8e-6 rms here until round 4: hipcc folded the final multiply into the f16 conversion separately for the hi store and
    for the lo subtraction, and the two roundings disagreed on near-ties (csrc/split_dev.h sp_value)."""
    g = torch.Generator().manual_seed(s + d)
    n, e = 3, heads * d
    qkv = torch.randn(n, s, 3 * e, generator=g)
    qkv[..., :e] *= 2.0                                              # scores up to ~ +-10
    scale = 1.0 / d ** 0.5
    q, k, v = (qkv[..., i * e:(i + 1) * e].double().view(n, s, heads, d).transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(n, s, e)
    dev = engine.device
    qd = qkv.to(dev)
    o32 = torch.full((n, s, e), float("nan"), device=dev)
    engine.attention(qd, o32, n, s, heads, d, scale, 0, 0)
    osp = torch.full((n, s, 2 * e), 0x7e00, dtype=torch.int16, device=dev)
    engine.attention(qd, osp, n, s, heads, d, scale, 0, 2)
    torch.cuda.synchronize()
    rms = ref.pow(2).mean().sqrt()
    for name, got in (("f32", o32.cpu().double()), ("x3", from_sp32(osp.cpu()).double())):
        err = got - ref
        rel, worst = (err.pow(2).mean().sqrt() / rms).item(), err.abs().max().item()
        print(f"attention s={s} heads={heads} d={d} {name}: rel rms {rel:.2e}, max|err| {worst:.2e}")
        assert rel < 1e-6 and worst < 4e-6, (name, rel, worst)


def test_attention_argument_errors(engine):
    from avcer_amd._lib import AvcerError

    x = torch.zeros(1, 300, 3 * 64, device=engine.device)
    with pytest.raises(AvcerError):
        engine.attention(x, x, 1, 300, 1, 64, 0.125, 0, 0)           # more than 256 tokens
    with pytest.raises(AvcerError):
        engine.attention(x, x, 1, 99, 1, 48, 0.125, 0, 0)            # head dimension
