"""The weights-direct form of the split-fp16 contraction (avcer_conv_gemm dtype 7 / 8, gemm.hip conv_gemm_wd_kernel)
against the LDS-staged form (dtype 5 / 6) it stands in for: the same product order per output element, so the results are
required to be BIT-IDENTICAL, on every gather path (plain matrix, padded 3x3 with border positions, strided 1x1,
un-padded multi-tap Conv1d with dilation, the channel-chunk-major K order, a second activation source), every epilogue
(scale / bias, residual before and after the activation, ReLU, GELU, sp32 and f32 output), and ragged M.
The staged form itself is checked against float64 in tests/test_gpu_kernels.py."""
import pytest
import torch

from avcer_amd.sp32 import from_sp32, to_sp32
from test_gpu_kernels import _desc

pytestmark = pytest.mark.gpu


def _both(engine, d, x, w, scale, bias, res, out_sp32, x2=None, tile_m=0):
    """Run dtype 5 / 6 and 7 / 8 on the same device tensors; returns the two raw output tensors."""
    d.tile_m = tile_m
    dev = engine.device
    xd = to_sp32(x).to(dev)
    x2d = None if x2 is None else to_sp32(x2).to(dev)
    wd = w.to(dev, torch.float32).contiguous()
    rows, frags = engine.split_weight_rows(wd), engine.weight_frags(wd)
    sd = None if scale is None else scale.to(dev)
    bd = None if bias is None else bias.to(dev)
    m = d.batch * d.out_h * d.out_w
    outs = []
    for dtype, wt in ((5 if out_sp32 else 6, rows), (7 if out_sp32 else 8, frags)):
        rd = None if res is None else (to_sp32(res).to(dev) if out_sp32 else res.to(dev).contiguous())
        y = torch.full((m, 2 * d.n) if out_sp32 else (m, d.n), -3, dtype=torch.int16 if out_sp32 else torch.float32, device=dev)
        if x2d is None:
            engine.conv_gemm(d, dtype, xd, wt, sd, bd, rd, y)
        else:
            engine.conv_gemm_dual(d, dtype, xd, x2d, wt, sd, bd, rd, y)
        torch.cuda.synchronize()
        outs.append(y.cpu())
    return outs


@pytest.mark.parametrize("m,k,n,act,out_sp32", [(300, 128, 256, 1, True), (129, 64, 512, 2, False), (1, 256, 256, 0, True),
                                                (5000, 1024, 768, 0, False), (40000, 64, 256, 1, True)])
@pytest.mark.parametrize("tile_m", [0, 112, 128])
def test_linear_bit_identical(engine, m, k, n, act, out_sp32, tile_m):
    g = torch.Generator().manual_seed(m + k + n)
    x, w = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5
    scale, bias = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g)
    d = _desc(batch=m, cin=k, x_stride_b=k, x_stride_h=k, x_stride_w=k, n=n, y_ld=n, r_ld=n, act=act, res_after_act=act == 2)
    a, b = _both(engine, d, x, w, scale, bias, res, out_sp32, tile_m=tile_m)
    assert torch.equal(a, b)
    got = from_sp32(b) if out_sp32 else b
    assert torch.isfinite(got).all() and got.abs().max() > 0.1


CONVS = [
    # b, h, w, c, kh, kw, stride, pad, dil, n, act
    (2, 9, 9, 64, 3, 3, 1, 1, 1, 256, 1),       # every position next to a border
    (3, 14, 14, 256, 3, 3, 1, 1, 1, 256, 1),    # stage 3 conv2 (channel-chunk-major K order: cin > 32, several taps)
    (3, 7, 7, 64, 1, 1, 2, 0, 1, 256, 0),       # strided 1x1
    (2, 28, 28, 128, 3, 3, 2, 1, 1, 256, 1),    # strided 3x3 (detector body)
    (2, 31, 1, 64, 6, 1, 3, 0, 2, 256, 0),      # Conv1d k6 s3 dil2 (the audio head's kind: un-padded, dilated, several taps)
    (2, 40, 1, 512, 2, 1, 2, 0, 1, 512, 2),     # Conv1d k2 s2 + gelu (feature extractor)
    (9, 55, 55, 64, 1, 1, 1, 0, 1, 256, 1),     # ragged M: 27225 positions
    # more 3x3 / stride 1 / pad 1 shapes (written for round 4's resident-patch experiment, profiles/experiments/README.md)
    (5, 14, 14, 256, 3, 3, 1, 1, 1, 512, 1),    # ragged M (980 positions), two n tiles, tiles that cross image boundaries
    (11, 14, 14, 512, 3, 3, 1, 1, 1, 256, 0),   # 16 channel chunks, no activation, 17 tiles
    (1, 14, 14, 64, 3, 3, 1, 1, 1, 256, 1),     # one image, two chunks: the shortest K loop
    (13, 7, 7, 512, 3, 3, 1, 1, 1, 512, 1),     # stage 4 conv2: 128 positions span four images
    (2, 28, 28, 128, 3, 3, 1, 1, 1, 256, 1),    # stage 2 geometry
]


@pytest.mark.parametrize("tile_m", [112, 128])
@pytest.mark.parametrize("out_sp32", [True, False])
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_bit_identical(engine, cfg, out_sp32, tile_m):
    b, h, w_, c, kh, kw, s, p, dil, n, act = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(b, h, w_, c, generator=g)
    w = torch.randn(n, kh * kw * c, generator=g) / (kh * kw * c) ** 0.5
    scale, bias = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g)
    sw, pw, dw = (1, 0, 1) if w_ == 1 else (s, p, dil)
    oh = (h + 2 * p - dil * (kh - 1) - 1) // s + 1
    ow = (w_ + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    res = torch.randn(b * oh * ow, n, generator=g)
    d = _desc(batch=b, in_h=h, in_w=w_, out_h=oh, out_w=ow, cin=c, kh=kh, kw=kw, stride_h=s, stride_w=sw, pad_h=p, pad_w=pw,
              dil_h=dil, dil_w=dw, x_stride_b=h * w_ * c, x_stride_h=w_ * c, x_stride_w=c, n=n, y_ld=n, r_ld=n, act=act)
    a, bb = _both(engine, d, x.reshape(-1, c), w, scale, bias, res, out_sp32, tile_m=tile_m)
    assert torch.equal(a, bb)


@pytest.mark.parametrize("tile_m", [112, 128])
def test_dual_source_bit_identical(engine, tile_m):
    """conv3 + downsample of a stage's first block in one contraction: K = [T2 (planes) | X at stride 2 (cin)]."""
    b, oh, planes, cin, n = 3, 14, 256, 512, 1024
    g = torch.Generator().manual_seed(7)
    t2 = torch.randn(b, oh, oh, planes, generator=g)
    xin = torch.randn(b, 2 * oh, 2 * oh, cin, generator=g)
    w = torch.randn(n, planes + cin, generator=g) / (planes + cin) ** 0.5
    bias = torch.randn(n, generator=g)
    d = _desc(batch=b, in_h=oh, in_w=oh, out_h=oh, out_w=oh, cin=planes, x_stride_b=oh * oh * planes, x_stride_h=oh * planes,
              x_stride_w=planes, n=n, y_ld=n, r_ld=n, act=1, x2_cin=cin, x2_stride=2, x2_stride_b=4 * oh * oh * cin,
              x2_stride_h=2 * oh * cin, x2_stride_w=cin)
    a, bb = _both(engine, d, t2.reshape(-1, planes), w, None, bias, None, True, x2=xin.reshape(-1, cin), tile_m=tile_m)
    assert torch.equal(a, bb)


# ----------------------------------------------------------------------------- the skinny form (dtype 9 / 10)
SKINNY_LINEAR = [(1, 128, 64, 0, True), (5, 96, 64, 1, False), (40, 416, 128, 2, True),   # K-steps: 4, 3, 13 (ring depths 12 / 8 / 4)
                 (49, 2048, 512, 1, True), (196, 256, 1024, 1, True), (199, 1024, 3072, 0, False),
                 (64, 512, 192, 3, False), (257, 4096, 1024, 0, False)]
SKINNY_CONVS = [
    # b, h, w, c, kh, kw, stride, pad, dil, n, act
    (1, 7, 7, 512, 3, 3, 1, 1, 1, 512, 1),      # stage 4 conv2 of ONE frame: 49 positions, K = 4608 (144 K-steps)
    (1, 14, 14, 256, 3, 3, 1, 1, 1, 256, 1),    # stage 3 conv2 of one frame: 196 positions, four row tiles
    (1, 14, 14, 256, 3, 3, 2, 1, 1, 256, 1),    # the strided conv2 of stage 3's last block
    (2, 9, 9, 128, 3, 3, 1, 1, 1, 64, 0),       # every position next to a border, two images in one row tile
    (1, 31, 1, 128, 4, 1, 3, 0, 2, 64, 0),      # un-padded Conv1d, dilated, several taps (the audio head's kind)
    (1, 28, 28, 128, 1, 1, 2, 0, 1, 128, 1),    # strided 1x1
]


def _skinny_pair(engine, d, x, w, scale, bias, res, out_sp32, tile_m=0):
    """dtype 5 / 6 (LDS-staged) and 9 / 10 (skinny, rows per wave tile chosen or forced) on the same device tensors."""
    dev = engine.device
    xd = to_sp32(x).to(dev)
    wd = w.to(dev, torch.float32).contiguous()
    rows, frags = engine.split_weight_rows(wd), engine.weight_frags(wd)
    m = d.batch * d.out_h * d.out_w
    outs = []
    for dtype, wt in ((5 if out_sp32 else 6, rows), (9 if out_sp32 else 10, frags)):
        rd = None if res is None else (to_sp32(res).to(dev) if out_sp32 else res.to(dev).contiguous())
        y = torch.full((m, 2 * d.n) if out_sp32 else (m, d.n), -3, dtype=torch.int16 if out_sp32 else torch.float32, device=dev)
        d.tile_m = tile_m if dtype >= 9 else 0
        engine.conv_gemm(d, dtype, xd, wt, None if scale is None else scale.to(dev), None if bias is None else bias.to(dev), rd, y)
        torch.cuda.synchronize()
        outs.append(y.cpu())
    return outs


@pytest.mark.parametrize("tile_m", [0, 16, 32, 64])
@pytest.mark.parametrize("m,k,n,act,out_sp32", SKINNY_LINEAR)
def test_skinny_linear_bit_identical(engine, m, k, n, act, out_sp32, tile_m):
    """The one-wave-per-tile form for a handful of positions (conv_gemm_skinny_kernel): the same bits as the tiled forms."""
    g = torch.Generator().manual_seed(m + k + n)
    x, w = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5
    scale, bias = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g)
    d = _desc(batch=m, cin=k, x_stride_b=k, x_stride_h=k, x_stride_w=k, n=n, y_ld=n, r_ld=n, act=act, res_after_act=act >= 2)
    a, b = _skinny_pair(engine, d, x, w, scale if m != 199 else None, bias if m != 64 else None, res if m != 1 else None, out_sp32, tile_m)
    assert torch.equal(a, b)
    got = from_sp32(b) if out_sp32 else b
    assert torch.isfinite(got).all() and got.abs().max() > 0.1


@pytest.mark.parametrize("tile_m", [16, 32, 64])
@pytest.mark.parametrize("out_sp32", [True, False])
@pytest.mark.parametrize("cfg", SKINNY_CONVS)
def test_skinny_conv_bit_identical(engine, cfg, out_sp32, tile_m):
    b, h, w_, c, kh, kw, s, p, dil, n, act = cfg
    g = torch.Generator().manual_seed(sum(cfg) + 1)
    x = torch.randn(b, h, w_, c, generator=g)
    w = torch.randn(n, kh * kw * c, generator=g) / (kh * kw * c) ** 0.5
    scale, bias = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g)
    sw, pw, dw = (1, 0, 1) if w_ == 1 else (s, p, dil)
    oh = (h + 2 * p - dil * (kh - 1) - 1) // s + 1
    ow = (w_ + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    res = torch.randn(b * oh * ow, n, generator=g)
    d = _desc(batch=b, in_h=h, in_w=w_, out_h=oh, out_w=ow, cin=c, kh=kh, kw=kw, stride_h=s, stride_w=sw, pad_h=p, pad_w=pw,
              dil_h=dil, dil_w=dw, x_stride_b=h * w_ * c, x_stride_h=w_ * c, x_stride_w=c, n=n, y_ld=n, r_ld=n, act=act)
    a, bb = _skinny_pair(engine, d, x.reshape(-1, c), w, scale, bias, res, out_sp32, tile_m)
    assert torch.equal(a, bb)


@pytest.mark.parametrize("out_sp32", [True, False])
def test_skinny_subsampled_residual_bit_identical(engine, out_sp32):
    """conv3 of a stage's last block at the even positions only: output row (b, oy, ox) adds residual row (b, 2 oy, 2 ox)."""
    b, oh, planes, n = 2, 7, 256, 1024
    g = torch.Generator().manual_seed(11)
    x, w = torch.randn(b * oh * oh, planes, generator=g), torch.randn(n, planes, generator=g) / planes ** 0.5
    scale, bias = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g)
    res = torch.randn(b * 4 * oh * oh, n, generator=g)
    d = _desc(batch=b, in_h=oh, in_w=oh, out_h=oh, out_w=oh, cin=planes, x_stride_b=oh * oh * planes, x_stride_h=oh * planes,
              x_stride_w=planes, n=n, y_ld=n, r_ld=n, act=1, r_sub=2, r_h=2 * oh, r_w=2 * oh)
    a, bb = _skinny_pair(engine, d, x, w, scale, bias, res, out_sp32)
    assert torch.equal(a, bb)
    ref = torch.relu(x.double() @ w.double().T * scale.double() + bias.double()
                     + res.double().reshape(b, 2 * oh, 2 * oh, n)[:, ::2, ::2].reshape(-1, n))
    got = (from_sp32(bb) if out_sp32 else bb).double()
    assert (got - ref).abs().max() < 2e-5 * ref.abs().max()


@pytest.mark.parametrize("tile_m", [0, 16, 64])
@pytest.mark.parametrize("out_sp32", [True, False])
def test_skinny_grouped_bit_identical(engine, out_sp32, tile_m):
    """The wav2vec2 pos-conv shape (groups of 64 channels, 16 taps, zero padding, GELU then residual) as one skinny launch:
    the fragment tiles of all groups side by side."""
    b, s, groups, cin, k = 2, 37, 4, 64, 16
    ctot = groups * cin
    g = torch.Generator().manual_seed(13)
    x = torch.randn(b * s, ctot, generator=g)
    w = torch.randn(ctot, k * cin, generator=g) / (k * cin) ** 0.5
    bias, res = torch.randn(ctot, generator=g), torch.randn(b * s, ctot, generator=g)
    d = _desc(batch=b, in_h=s, in_w=1, out_h=s, out_w=1, cin=cin, kh=k, kw=1, pad_h=k // 2, x_stride_b=s * ctot,
              x_stride_h=ctot, x_stride_w=ctot, n=64, y_ld=ctot, r_ld=ctot, act=2, res_after_act=1, groups=groups)
    dev = engine.device
    xd, wd = to_sp32(x).to(dev), w.to(dev).contiguous()
    rows, frags = engine.split_weight_rows(wd), engine.weight_frags(wd)
    outs = []
    for dtype, wt in ((5 if out_sp32 else 6, rows), (9 if out_sp32 else 10, frags)):
        rd = to_sp32(res).to(dev) if out_sp32 else res.to(dev)
        y = torch.full((b * s, 2 * ctot) if out_sp32 else (b * s, ctot), -3, dtype=torch.int16 if out_sp32 else torch.float32, device=dev)
        d.tile_m = tile_m if dtype >= 9 else 0
        engine.conv_gemm(d, dtype, xd, wt, None, bias.to(dev), rd, y)
        torch.cuda.synchronize()
        outs.append(y.cpu())
    assert torch.equal(outs[0], outs[1])
    got = from_sp32(outs[1]) if out_sp32 else outs[1]
    assert torch.isfinite(got).all() and got.abs().max() > 0.1


def test_skinny_form_refuses_what_it_cannot_do(engine):
    from avcer_amd._lib import AvcerError

    dev = engine.device
    x = to_sp32(torch.randn(4160, 64)).to(dev)
    y = torch.zeros(4160, 64, device=dev)
    w = torch.zeros(64 * 64 * 2 + 128, dtype=torch.int16, device=dev)
    d = _desc(batch=4160, cin=64, x_stride_b=64, x_stride_h=64, x_stride_w=64, n=64, y_ld=64, r_ld=64)   # M > 4096
    with pytest.raises(AvcerError, match="skinny form"):
        engine.conv_gemm(d, 10, x, w, None, None, None, y)
    d = _desc(batch=64, cin=64, x_stride_b=64, x_stride_h=64, x_stride_w=64, n=64, y_ld=64, r_ld=64)
    d.tile_m = 128
    with pytest.raises(AvcerError, match="tile_m"):
        engine.conv_gemm(d, 10, x, w, None, None, None, y)


def test_shapes_outside_the_form_are_refused(engine):
    from avcer_amd._lib import AvcerError

    dev = engine.device
    x = to_sp32(torch.randn(64, 96)).to(dev)
    y = torch.zeros(64, 256, device=dev)
    w = torch.zeros(256 * 96 * 2, dtype=torch.int16, device=dev)
    d = _desc(batch=64, cin=96, x_stride_b=96, x_stride_h=96, x_stride_w=96, n=256, y_ld=256, r_ld=256)   # 3 K-steps: odd
    with pytest.raises(AvcerError, match="even number of K-steps"):
        engine.conv_gemm(d, 8, x, w, None, None, None, y)
    d = _desc(batch=64, cin=64, x_stride_b=64, x_stride_h=64, x_stride_w=64, n=128, y_ld=128, r_ld=128)   # N = 128
    with pytest.raises(AvcerError, match="N % 256"):
        engine.conv_gemm(d, 8, x, w, None, None, None, y)
    d = _desc(batch=64, cin=64, x_stride_b=64, x_stride_h=64, x_stride_w=64, n=256, y_ld=256, r_ld=256, tile_m=96)
    with pytest.raises(AvcerError, match="tile_m"):
        engine.conv_gemm(d, 8, x, w, None, None, None, y)
