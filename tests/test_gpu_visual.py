"""GPU parity of the static CNN, the LSTM and the per-frame harness against the oracle / golden vectors."""
import numpy as np
import pytest
import torch

from avcer_amd import synth, video_pipeline
from avcer_amd.engine import MODE_BF16, MODE_F16X3, MODE_FP32
from avcer_amd.sp32 import raw_to_f32
from oracle import video as ov

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-4  # north_star: probabilities within 1e-4 of the CPU reference in fp32 mode


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _stage_out(taps, name):
    """What the library holds as the output of a stage: stages 1-3 only at the positions (2 oy, 2 ox) that the next stage's
    stride-2 1x1 convolutions read (video.py:12-19,140-149) -- the other positions of the reference's tensor are never used."""
    t = _nhwc(taps[name])
    return t if name == "layer4" else t[:, ::2, ::2].contiguous()


def test_static_stage_taps_fp32(engine_static, sd_static):
    frames = synth.face_frames(1234, 8)
    taps = {}
    with torch.no_grad():
        x = ov.pth_processing(frames)
        ov.resnet50_forward(sd_static, x, taps)
        p = "layer1.0"
        c1 = torch.relu(ov._bn(torch.nn.functional.conv2d(taps["stem"], sd_static[p + ".conv1.weight"]), sd_static, p + ".batch_norm1"))
    refs = {"stem": _nhwc(taps["stem"]), "l1b0_c1": _nhwc(c1), "layer1": _stage_out(taps, "layer1"),
            "layer2": _stage_out(taps, "layer2"), "layer3": _stage_out(taps, "layer3"), "layer4": _stage_out(taps, "layer4"),
            "avgpool": taps["avgpool"]}
    pre = torch.zeros(8, 230, 230, 4)
    pre[:, 2:226, 2:226, :3] = _nhwc(x)
    refs = {"pre": pre, **refs}
    report = []
    for name, ref in refs.items():
        dst = engine_static.debug_tap(name, ref.numel())
        engine_static.static_forward(torch.from_numpy(frames), MODE_FP32)
        torch.cuda.synchronize()
        assert engine_static.debug_tap_copied() == ref.numel() * 4, name
        got = dst.cpu().view(ref.shape)
        err = (got - ref).abs().max().item()
        report.append((name, err, ref.abs().max().item()))
    print("static fp32 stage errors (name, max|err|, max|ref|):", report)
    for name, err, mx in report:
        assert err < 2e-4 * max(mx, 1.0), report


def _sp32_to_f32(raw_i16: torch.Tensor, shape):
    """Decode an sp32 tensor (per 32 channels: 32 fp16 hi then 32 fp16 lo; value = hi + lo) tapped as raw int16."""
    return raw_to_f32(raw_i16, shape)


def test_static_stage_taps_x3(engine_static, sd_static):
    """Split-bf16 mode, stage by stage: the fused stem + max-pool kernel and the fused bottleneck chains of stages 1-2
    (csrc/fused.hip) against the oracle's taps (video.py:98-117, 43-60).  9 frames: 9*55*55 is not a multiple of 128."""
    frames = synth.face_frames(4321, 9)
    taps = {}
    with torch.no_grad():
        ov.resnet50_forward(sd_static, ov.pth_processing(frames), taps)
    report = []
    for name in ("stem", "layer1", "layer2", "layer3", "layer4"):
        ref = _nhwc(taps[name]) if name == "stem" else _stage_out(taps, name)
        dst = engine_static.debug_tap(name, ref.numel() * 2, dtype=torch.int16)
        engine_static.static_forward(torch.from_numpy(frames), MODE_F16X3)
        torch.cuda.synchronize()
        assert engine_static.debug_tap_copied() == ref.numel() * 4, name
        got = _sp32_to_f32(dst.cpu(), ref.shape)
        report.append((name, (got - ref).abs().max().item(), ref.abs().max().item()))
    print("static x3 stage errors (name, max|err|, max|ref|):", report)
    for name, err, mx in report:
        assert err < 3e-4 * max(mx, 1.0), report


def test_static_matches_golden_and_oracle_fp32(engine_static, sd_static, golden):
    g = golden("static")
    frames = synth.face_frames(1234, 8)
    logits, probs, feats = [t.cpu().numpy() for t in engine_static.static_forward(torch.from_numpy(frames), MODE_FP32)]
    assert np.abs(probs - g["probs"]).max() < PROB_TOL
    assert np.abs(logits - g["logits"]).max() < 1e-4  # measured 9e-6
    assert np.abs(feats - g["feats"]).max() < 1e-4
    assert (probs.argmax(1) == g["probs"].argmax(1)).all()
    print("static fp32 max|dprob|", np.abs(probs - g["probs"]).max(), "max|dlogit|", np.abs(logits - g["logits"]).max())
    # the preprocessed-tensor entry point (exact argument of pth_model_static) gives the same numbers
    l2, p2, f2 = engine_static.static_forward_nchw(ov.pth_processing(frames), MODE_FP32)
    assert torch.equal(l2.cpu(), torch.from_numpy(logits)) and torch.equal(f2.cpu(), torch.from_numpy(feats))


def test_static_split_bf16_meets_parity_gate(engine_static, golden):
    """AVCER_MODE_F16X3 (bf16 MFMA on hi/lo-split operands) must satisfy the same 1e-4 gate as the f32 mode."""
    g = golden("static")
    frames = synth.face_frames(1234, 8)
    logits, probs, feats = [t.cpu().numpy() for t in engine_static.static_forward(torch.from_numpy(frames), MODE_F16X3)]
    print("static split-fp16 max|dprob|", np.abs(probs - g["probs"]).max(), "max|dlogit|", np.abs(logits - g["logits"]).max())
    assert np.abs(probs - g["probs"]).max() < PROB_TOL
    assert (probs.argmax(1) == g["probs"].argmax(1)).all()


def test_static_bf16_reports_and_keeps_argmax(engine_static, golden):
    g = golden("static")
    frames = synth.face_frames(1234, 8)
    _, probs, _ = [t.cpu().numpy() for t in engine_static.static_forward(torch.from_numpy(frames), MODE_BF16)]
    d = np.abs(probs - g["probs"]).max()
    print("static bf16 max|dprob|", d)
    assert np.isfinite(probs).all() and d < 5e-2
    top2 = np.sort(g["probs"], axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2 * d
    assert (probs.argmax(1)[clear] == g["probs"].argmax(1)[clear]).all()


def test_static_resize_path(engine_static, sd_static):
    odd = synth.u8(77, "odd4", (3, 150, 131, 3))
    with torch.no_grad():
        x = ov.pth_processing(np.stack([ov.nearest_resize_u8(f) for f in odd]))
        lg, _ = ov.resnet50_forward(sd_static, x)
        ref = torch.softmax(lg, 1).numpy()
    # the f32 mode resizes in the preprocessing launch, the split-fp16 mode inside the fused stem (fused.hip stem_pool_u8_kernel)
    for mode in (MODE_FP32, MODE_F16X3):
        _, probs, _ = engine_static.static_forward(torch.from_numpy(odd), mode)
        assert np.abs(probs.cpu().numpy() - ref).max() < PROB_TOL, mode


def test_stage3_tail_ragged_rows_match_whole_batch(engine_static):
    """The stage-3 tail kernel (conv3 + residual + next conv1, fused.hip) on a position count that is not a multiple of its
    128-row tile: 85 frames = 16660 positions = 130 tiles + 20 rows (above api.hip kTailPairRows, where the fused kernel
    serves).  The rows of the ragged last tile are bit-identical to the same frames inside a larger call (rows past M are
    dropped by the store descriptor, never written elsewhere)."""
    frames = torch.from_numpy(synth.face_frames(5, 100))
    big = [t.cpu() for t in engine_static.static_forward(frames, MODE_F16X3)]
    part = [t.cpu() for t in engine_static.static_forward(frames[:85], MODE_F16X3)]
    assert all(torch.equal(a[:85], b) for a, b in zip(big, part))


def test_static_one_frame_matches_its_rows_in_a_batch_x3(engine_static):
    """A frame per call (what the drop-in mirror hands over) selects other kernels than a batch does -- the skinny
    contraction (conv_gemm dtype 9 / 10) wherever a layer's grid of wave tiles fits the chip, the stage-3 tails as two
    contractions instead of the fused kernel (api.hip kTailPairRows), fc1 on 16-row tiles: bit-identical features,
    probabilities and logits for 1, 2, 5 and 40 frames against the same frames inside a call of 100."""
    frames = torch.from_numpy(synth.face_frames(17, 100))
    big = [t.cpu() for t in engine_static.static_forward(frames, MODE_F16X3)]
    for lo, hi in ((0, 1), (7, 9), (33, 38), (60, 100)):
        part = [t.cpu() for t in engine_static.static_forward(frames[lo:hi], MODE_F16X3)]
        assert all(torch.equal(a[lo:hi], b) for a, b in zip(big, part)), (lo, hi)


def test_static_batch_invariance_256(engine_static):
    """BASELINE config 2 size: results must not depend on batch composition (sub-batching at 256)."""
    frames = torch.from_numpy(synth.face_frames(99, 300))
    big = [t.cpu() for t in engine_static.static_forward(frames, MODE_FP32)]
    small = [t.cpu() for t in engine_static.static_forward(frames[250:262], MODE_FP32)]
    for a, b in zip(big, small):
        assert torch.equal(a[250:262], b)
    again = [t.cpu() for t in engine_static.static_forward(frames, MODE_FP32)]
    assert all(torch.equal(a, b) for a, b in zip(big, again))
    p = big[1]
    assert torch.isfinite(p).all() and (p.sum(1) - 1).abs().max() < 1e-5


def test_lstm_matches_golden_and_oracle(engine_dynamic, sd_dynamic, golden):
    w = np.maximum(synth.centered(5, "lstm_in", (4, 10, 512), 1.0), 0).astype(np.float32)
    w[0] = w[0, 0]
    out = engine_dynamic.dynamic_forward(torch.from_numpy(w)).cpu().numpy()
    assert np.abs(out - golden("lstm")["logits"]).max() < 2e-5
    big = np.maximum(synth.centered(6, "lstm_in2", (333, 10, 512), 1.5), 0).astype(np.float32)
    with torch.no_grad():
        ref = ov.lstm_forward(sd_dynamic, torch.from_numpy(big)).numpy()
    out = engine_dynamic.dynamic_forward(torch.from_numpy(big)).cpu().numpy()
    print("lstm max|dlogit|", np.abs(out - ref).max())
    assert np.abs(out - ref).max() < 3e-6  # measured 2.7e-7


def test_visual_harness_matches_reference_tables(engine_static, engine_dynamic, golden):
    g = golden("visual_harness")
    clip = torch.from_numpy(synth.face_frames(4321, 16))
    for name in ("gap25", "gap30", "lead25", "full25"):
        st, dy = video_pipeline.visual_forward(engine_static, clip, g[f"{name}_present"], float(g[f"{name}_fps"]))
        assert np.abs(st.cpu().numpy() - g[f"{name}_static"]).max() < PROB_TOL, name
        print("harness", name, "max|d dynamic logit|", np.abs(dy.cpu().numpy() - g[f"{name}_dynamic"]).max())
        assert np.abs(dy.cpu().numpy() - g[f"{name}_dynamic"]).max() < 1e-5, name  # measured 9e-7
    # batched over clips == clip by clip
    clips = torch.stack([clip, torch.from_numpy(synth.face_frames(4322, 16))])
    present = np.stack([g["gap25_present"], g["lead25_present"]])
    st, dy = video_pipeline.visual_forward(engine_static, clips, present, 25)
    s0, d0 = video_pipeline.visual_forward(engine_static, clips[1], present[1], 25)
    assert torch.equal(st[1], s0) and torch.equal(dy[1], d0)


def test_lstm_split_bf16_mode(engine, sd_dynamic, golden):
    """x3 projections (f32 state): still f32-grade against the reference LSTM."""
    from avcer_amd.engine import MODE_F16X3
    engine.load_dynamic(sd_dynamic)
    w = np.maximum(synth.centered(5, "lstm_in", (4, 10, 512), 1.0), 0).astype(np.float32)
    w[0] = w[0, 0]
    out = engine.dynamic_forward(torch.from_numpy(w), MODE_F16X3).cpu().numpy()
    d = np.abs(out - golden("lstm")["logits"]).max()
    print("lstm x3 max|dlogit|", d)
    assert d < 2e-5  # measured 1.8e-6


def test_lstm_one_window_matches_its_row_in_a_batch_x3(engine, sd_dynamic):
    """A window per call (the drop-in mirror) runs its recurrent contractions on the skinny form, a few thousand windows on the
    tiled ones: the same bits for the same window either way."""
    from avcer_amd.engine import MODE_F16X3
    engine.load_dynamic(sd_dynamic)
    g = torch.Generator().manual_seed(3)
    w = torch.relu(torch.randn(3000, 10, 512, generator=g))
    big = engine.dynamic_forward(w, MODE_F16X3).cpu()
    for lo, hi in ((0, 1), (17, 22), (2000, 2300)):
        part = engine.dynamic_forward(w[lo:hi], MODE_F16X3).cpu()
        assert torch.equal(big[lo:hi], part), (lo, hi)
    assert torch.isfinite(big).all()


def test_kernel_families_follow_the_batch(engine_static):
    """avcer_profile_read_families (the bench line's roofline.per_family): which kernel family serves the static CNN is a function
    of the batch -- one frame: every contraction with a small grid on the skinny form, the stage-3 tails as two contractions (no
    fused tail launch); 100 frames: the four fused tails, the chains, the weights-direct form -- and the event times, FLOPs and
    bytes of every family that launched are positive."""
    frames = torch.from_numpy(synth.face_frames(3, 100))
    engine_static.static_forward(frames[:2], MODE_F16X3)   # weights split, workspaces sized
    torch.cuda.synchronize()
    got = {}
    for n in (1, 100):
        engine_static.profile_enable(True)
        engine_static.static_forward(frames[:n], MODE_F16X3)
        got[n] = engine_static.profile_read_families()
        engine_static.profile_enable(False)
    one, many = got[1], got[100]
    assert one["bneck_tail2_kernel"][1] == 0 and many["bneck_tail2_kernel"][1] == 4
    assert one["conv_gemm_skinny_kernel"][1] >= 25 and many["conv_gemm_skinny_kernel"][1] <= 2
    assert one["conv_gemm_wd_kernel"][1] == 0 and many["conv_gemm_wd_kernel"][1] >= 8
    assert one["bneck_kernel"][1] == many["bneck_kernel"][1] == 6 and one["stem_pool_kernel"][1] == 1
    for fam in (one, many):
        for name, (ms, launches, flops, nbytes) in fam.items():
            assert (ms > 0 and flops > 0 and nbytes > 0) if launches else (ms == 0 and flops == 0), name
    # the same graph either way: FLOPs per frame agree to the padding of the skinny tiles' own accounting (none: algorithmic)
    f1, f100 = sum(v[2] for v in one.values()), sum(v[2] for v in many.values())
    assert abs(f100 / 100 - f1) < 1e-6 * f1


def test_profile_launch_log_matches_the_family_sums(engine_static):
    """avcer_profile_read_launches (tools/wd_traffic.py matches it with a rocprofv3 counter pass): one entry per MFMA launch in
    launch order with its family, event time, algorithmic FLOPs, compulsory bytes and the contraction's M, N, K -- the same
    launches and the same sums the per-family read reports for the same call."""
    frames = torch.from_numpy(synth.face_frames(3, 100))
    engine_static.static_forward(frames, MODE_F16X3)
    torch.cuda.synchronize()
    engine_static.profile_enable(True)
    engine_static.static_forward(frames, MODE_F16X3)
    fams = engine_static.profile_read_families()
    engine_static.static_forward(frames, MODE_F16X3)
    log = engine_static.profile_read_launches()
    engine_static.profile_enable(False)
    assert len(log) == sum(v[1] for v in fams.values()) and log[0]["family"] == "stem_pool_kernel"
    for name, (ms, launches, flops, nbytes) in fams.items():
        mine = [l for l in log if l["family"] == name]
        assert len(mine) == launches
        assert abs(sum(l["flops"] for l in mine) - flops) <= 1e-9 * max(flops, 1) and abs(sum(l["bytes"] for l in mine) - nbytes) <= 1e-9 * max(nbytes, 1)
    gemms = [l for l in log if l["family"].startswith("conv_gemm")]
    assert all(l["ms"] > 0 for l in log) and all(abs(2.0 * l["m"] * l["n"] * l["k"] - l["flops"]) <= 1e-9 * l["flops"] for l in gemms)
    # layer3's first 3 x 3 convolution of the 100 frames: 19600 positions, 256 channels out, K = 9 x 256
    assert any((l["m"], l["n"], l["k"]) == (100 * 14 * 14, 256, 2304) for l in gemms)
