"""state_dict -> packed weight blob for libavcer_hip.so.

Accepts the reference's own state_dict key names, so a real checkpoint
(`torch.load("FER_static_ResNet50_AffectNet.pt")`, the LSTM file, `epoch_63.pth["model_state_dict"]`) packs the
same way as the synthetic ones.  Layout transforms done here (all in float32 numpy):
  * conv weights OIHW -> [O][kh][kw][I] (K contiguous, matches the NHWC gather order of the kernel);
  * inference BatchNorm folded into per-channel (scale, bias) applied in the GEMM epilogue;
  * the 7x7 stem packed as 8 tap rows x (8 pixels x 4 channels) with zero taps (see kernels.hip preprocess);
  * conv3 and the downsample conv of the first block of every ResNet stage concatenated along K (one dual-source GEMM);
  * LSTM bias_ih + bias_hh summed; q/k/v projection matrices concatenated into one [3E, E] GEMM;
  * wav2vec2 positional-conv weight-norm materialised (w = g * v / ||v||, norm over dims 0,1) and split per group.

Blob: "AVCERW01", u32 count, u32 0, then `count` entries {char name[96]; u32 ndim; u32 0; i64 dims[4];
u64 offset; u64 nbytes}, then 64-byte aligned little-endian float32 payloads.
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

STATIC_BN_EPS = 1e-3   # architectures/video.py:21 (eps=0.001 on every BatchNorm2d)
AUDIO_BN_EPS = 1e-5    # torch.nn.BatchNorm1d default, architectures/audio_8_cl.py:150,154
RESNET_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))
PE_ROWS = 256          # attention kernel limit: S <= 256 tokens


def _np(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def _f32(v) -> np.ndarray:
    return np.ascontiguousarray(_np(v), dtype=np.float32)


def _unwrap(sd):
    """The forms a checkpoint file comes in: the audio trainer saves {"epoch": ..., "model_state_dict": ...}
    (audio/net_trainer/net_trainer.py:273-285, read back at get_prob_audio_8_cl.py:58-65), and a model saved from
    nn.DataParallel prefixes every key with "module." (stripped by retina_face_predictor.py:28-33 for the detector)."""
    if "model_state_dict" in sd and not hasattr(sd["model_state_dict"], "shape"):
        sd = sd["model_state_dict"]
    if sd and all(k.startswith("module.") for k in sd):
        sd = {k[len("module."):]: v for k, v in sd.items()}
    return sd


def _bn_fold(sd, p, eps, conv_bias=None):
    g, b = _f32(sd[p + ".weight"]), _f32(sd[p + ".bias"])
    m, v = _f32(sd[p + ".running_mean"]), _f32(sd[p + ".running_var"])
    s = (g / np.sqrt(v + np.float32(eps))).astype(np.float32)
    shift = -m if conv_bias is None else (_f32(conv_bias) - m)
    return s, (shift * s + b).astype(np.float32)


def _conv_w(w) -> np.ndarray:
    w = _f32(w)  # [O, I, kh, kw] -> [O, kh*kw*I]
    return np.ascontiguousarray(w.transpose(0, 2, 3, 1)).reshape(w.shape[0], -1)


def _conv1d_w(w) -> np.ndarray:
    w = _f32(w)  # [O, I, k] -> [O, k*I]
    return np.ascontiguousarray(w.transpose(0, 2, 1)).reshape(w.shape[0], -1)


PIXEL_MEANS = (91.4953, 103.8827, 131.0912)  # data/utils.py:36-38, in the channel order the network is fed (B, G, R)


def stem_border_shifts(w, scale, shift) -> np.ndarray:
    """Shifts of the fused stem when it contracts RAW pixel values (fused.hip stem_pool_kernel<true>).

    The reference convolves the normalised image p - mu, zero-padded 2 before / 3 after (video.py:68-80), so a stem position sums
    w (p - mu) over the taps INSIDE the image: sum_valid w p - sum_valid w mu.  The second term depends on the output channel and
    on which taps are valid -- rows: position 0 loses taps 0, 1; position 110 loses tap 6 (position 111 is never pooled);
    columns alike -- i.e. on one of 9 border classes 3 * row_class + col_class (0 first, 1 interior, 2 last).  Returns f32
    [9, 64]: BN shift - BN scale * sum_valid w mu, accumulated in float64."""
    w = np.asarray(w, np.float64)                      # [64, 3, 7, 7]
    mu = np.asarray(PIXEL_MEANS, np.float64)
    valid = (slice(2, 7), slice(0, 7), slice(0, 6))
    out = np.empty((9, w.shape[0]), np.float64)
    for ry, vy in enumerate(valid):
        for rx, vx in enumerate(valid):
            c = np.einsum("ocyx,c->o", w[:, :, vy, vx], mu)
            out[3 * ry + rx] = np.asarray(shift, np.float64) - np.asarray(scale, np.float64) * c
    return out.astype(np.float32)


def _fold_fused_weights(out, li: int, blocks: int) -> None:
    """`*.wf` = weight rows times the BN scale, for the layers the x3 mode runs inside fused kernels (csrc/fused.hip), whose
    epilogues only add the BN shift.  Shared by the static CNN and the detector body: a bottleneck WITHOUT spatial stride has the
    same shape in both (video.py:43-60 puts a stage's stride on conv1 of its first block, torchvision on conv2 of the same block)."""
    if li <= 2:  # stages that run as fused chains conv2 -> conv3 (+x) -> next conv1:
        # stage 1 from its first block (stride 1; conv3 + downsample = c3d), stage 2 from its second block.
        for b in range(0 if li == 1 else 1, blocks):
            head = b == (0 if li == 1 else 1)  # conv1 of the chain's first block runs on its own (plain conv_gemm)
            for i in (1, 2, 3):
                if (i == 1 and head) or (i == 3 and b == 0):
                    continue
                k = f"l{li}.{b}.c{i}"
                out[k + ".wf"] = np.ascontiguousarray(out[k + ".w"] * out[k + ".s"][:, None])
    if li == 3:  # stage 3: conv3 (+x) of block b and conv1 of block b+1 share a launch (bneck_tail2_kernel), b = 1..blocks-2
        for b in range(1, blocks - 1):
            for k in (f"l{li}.{b}.c3", f"l{li}.{b + 1}.c1"):
                out[k + ".wf"] = np.ascontiguousarray(out[k + ".w"] * out[k + ".s"][:, None])


def pack_static(sd) -> "OrderedDict[str, np.ndarray]":
    """ResNet50(7) state_dict (architectures/video.py:93-166)."""
    sd = _unwrap(sd)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    w = _f32(sd["conv_layer_s2_same.weight"])  # [64, 3, 7, 7]
    stem = np.zeros((64, 8, 8, 4), np.float32)
    stem[:, :7, :7, :3] = w.transpose(0, 2, 3, 1)
    out["stem.w"] = stem.reshape(64, 256)
    out["stem7.w"] = np.ascontiguousarray(stem[:, :7].reshape(64, 224))  # 7 tap rows only: stem_pool_kernel (x3 mode)
    out["stem.s"], out["stem.b"] = _bn_fold(sd, "batch_norm1", STATIC_BN_EPS)
    out["stem.b9"] = stem_border_shifts(w, out["stem.s"], out["stem.b"])
    for li, (planes, blocks, _) in enumerate(RESNET_STAGES, start=1):
        for b in range(blocks):
            src, dst = f"layer{li}.{b}", f"l{li}.{b}"
            for i in (1, 2, 3):
                out[f"{dst}.c{i}.w"] = _conv_w(sd[f"{src}.conv{i}.weight"])
                out[f"{dst}.c{i}.s"], out[f"{dst}.c{i}.b"] = _bn_fold(sd, f"{src}.batch_norm{i}", STATIC_BN_EPS)
            if b == 0:
                # relu(bn3(conv3(t2)) + bn_d(conv_d(x))) as ONE contraction over K = [t2 | x strided]: the two BN scales
                # are folded into the weight rows (scale 1 in the epilogue), the two BN shifts are summed
                wd = _conv_w(sd[f"{src}.i_downsample.0.weight"])
                s_d, b_d = _bn_fold(sd, f"{src}.i_downsample.1", STATIC_BN_EPS)
                w3, s3, b3 = out.pop(f"{dst}.c3.w"), out.pop(f"{dst}.c3.s"), out.pop(f"{dst}.c3.b")
                out[f"{dst}.c3d.w"] = np.ascontiguousarray(np.concatenate([w3 * s3[:, None], wd * s_d[:, None]], axis=1))
                out[f"{dst}.c3d.b"] = (b3 + b_d).astype(np.float32)
        _fold_fused_weights(out, li, blocks)
    out["fc1.w"], out["fc1.b"] = _f32(sd["fc1.weight"]), _f32(sd["fc1.bias"])
    out["fc2.w"], out["fc2.b"] = _f32(sd["fc2.weight"]), _f32(sd["fc2.bias"])
    return out


def permute_rows_for_mfma(w: np.ndarray) -> np.ndarray:
    """Row order of every split-fp16 weight matrix on the device (the library applies it when it builds the split copies,
    csrc/kernels.hip split_weight_rows_kernel; this numpy twin exists for tests): inside every group of 32 output
    channels, stored row 16t + 4g + r holds channel 8g + 4t + r (t = 0,1; g = 0..3; r = 0..3), so that the two
    16-row MFMA tiles of a group leave every lane group g with the 8 consecutive channels 8g..8g+7."""
    n = w.shape[0]
    assert n % 32 == 0
    i = np.arange(32)
    t, g, r = i // 16, (i % 16) // 4, i % 4
    src = 8 * g + 4 * t + r
    idx = (np.arange(0, n, 32)[:, None] + src[None, :]).reshape(-1)
    return np.ascontiguousarray(w[idx])


def pack_face(sd) -> "OrderedDict[str, np.ndarray]":
    """RetinaFace(cfg_re50).state_dict() (retina_face/retina_face.py:46-76; `body.*` = torchvision ResNet-50 children)."""
    eps = 1e-5  # torch.nn.BatchNorm2d default, used by torchvision's ResNet and by retina_face_net.py
    sd = _unwrap(sd)
    sd = {(k.split("module.", 1)[-1] if k.startswith("module.") else k): v for k, v in sd.items()}  # predictor.py:28-33
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    w = _f32(sd["body.conv1.weight"])  # [64, 3, 7, 7] -> 8 tap rows x (8 pixels x 4 channels), zero-filled
    stem = np.zeros((64, 8, 8, 4), np.float32)
    stem[:, :7, :7, :3] = w.transpose(0, 2, 3, 1)
    out["stem.w"] = stem.reshape(64, 256)
    out["stem7.w"] = np.ascontiguousarray(stem[:, :7].reshape(64, 224))  # 7 tap rows only: the fused stem of the x3 mode
    out["stem.s"], out["stem.b"] = _bn_fold(sd, "body.bn1", eps)
    for li, (planes, blocks, _) in enumerate(RESNET_STAGES, start=1):
        for b in range(blocks):
            src, dst = f"body.layer{li}.{b}", f"l{li}.{b}"
            for i in (1, 2, 3):
                out[f"{dst}.c{i}.w"] = _conv_w(sd[f"{src}.conv{i}.weight"])
                out[f"{dst}.c{i}.s"], out[f"{dst}.c{i}.b"] = _bn_fold(sd, f"{src}.bn{i}", eps)
            if b == 0:  # conv3 + downsample as one contraction over K = [t2 | x strided], as in pack_static
                wd = _conv_w(sd[f"{src}.downsample.0.weight"])
                s_d, b_d = _bn_fold(sd, f"{src}.downsample.1", eps)
                w3, s3, b3 = out.pop(f"{dst}.c3.w"), out.pop(f"{dst}.c3.s"), out.pop(f"{dst}.c3.b")
                out[f"{dst}.c3d.w"] = np.ascontiguousarray(np.concatenate([w3 * s3[:, None], wd * s_d[:, None]], axis=1))
                out[f"{dst}.c3d.b"] = (b3 + b_d).astype(np.float32)
        _fold_fused_weights(out, li, blocks)  # the body's stride-1 bottlenecks run on the same fused kernels in the x3 mode
    for i in (1, 2, 3):
        out[f"fpn.o{i}.w"] = _conv_w(sd[f"fpn.output{i}.0.weight"])
        out[f"fpn.o{i}.s"], out[f"fpn.o{i}.b"] = _bn_fold(sd, f"fpn.output{i}.1", eps)
    for i in (1, 2):
        out[f"fpn.m{i}.w"] = _conv_w(sd[f"fpn.merge{i}.0.weight"])
        out[f"fpn.m{i}.s"], out[f"fpn.m{i}.b"] = _bn_fold(sd, f"fpn.merge{i}.1", eps)
    for i in (1, 2, 3):
        for src, dst in (("conv3X3", "c3"), ("conv5X5_1", "c51"), ("conv5X5_2", "c52"), ("conv7X7_2", "c72"),
                         ("conv7x7_3", "c73")):
            out[f"ssh{i}.{dst}.w"] = _conv_w(sd[f"ssh{i}.{src}.0.weight"])
            out[f"ssh{i}.{dst}.s"], out[f"ssh{i}.{dst}.b"] = _bn_fold(sd, f"ssh{i}.{src}.1", eps)
    for i in range(3):  # the three 1x1 heads of a level as one [64, 256] GEMM: class 0-3, bbox 4-11, landmarks 12-31
        wh, bh = np.zeros((64, 256), np.float32), np.zeros((64,), np.float32)
        row = 0
        for head, nrow in (("ClassHead", 4), ("BboxHead", 8), ("LandmarkHead", 20)):
            wh[row:row + nrow] = _f32(sd[f"{head}.{i}.conv1x1.weight"]).reshape(nrow, 256)
            bh[row:row + nrow] = _f32(sd[f"{head}.{i}.conv1x1.bias"])
            row += nrow
        out[f"head{i}.w"], out[f"head{i}.b"] = wh, bh
    return out


def pack_dynamic(sd) -> "OrderedDict[str, np.ndarray]":
    """LSTMPyTorch state_dict (architectures/video.py:169-185)."""
    sd = _unwrap(sd)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name in ("lstm1", "lstm2"):
        out[f"{name}.wih.w"] = _f32(sd[f"{name}.weight_ih_l0"])
        out[f"{name}.whh.w"] = _f32(sd[f"{name}.weight_hh_l0"])
        out[f"{name}.b"] = (_f32(sd[f"{name}.bias_ih_l0"]) + _f32(sd[f"{name}.bias_hh_l0"])).astype(np.float32)
    out["fc.w"], out["fc.b"] = _f32(sd["fc.weight"]), _f32(sd["fc.bias"])
    return out


def pos_conv_weight(sd) -> np.ndarray:
    p = "wav2vec2.encoder.pos_conv_embed.conv"
    if p + ".parametrizations.weight.original0" in sd:
        g, v = _f32(sd[p + ".parametrizations.weight.original0"]), _f32(sd[p + ".parametrizations.weight.original1"])
    elif p + ".weight_g" in sd:
        g, v = _f32(sd[p + ".weight_g"]), _f32(sd[p + ".weight_v"])
    else:
        return _f32(sd[p + ".weight"])
    import torch

    # the very function torch's weight_norm parametrisation evaluates (dim=2), so the packed weight is bit-identical
    return torch._weight_norm(torch.from_numpy(v), torch.from_numpy(g), 2).numpy()


def pack_audio(sd) -> "OrderedDict[str, np.ndarray]":
    """ExprModelV3 / ExprModelV2 state_dict (architectures/audio_8_cl.py:131-190, audio_7_cl.py), bare or inside the
    trainer's {"model_state_dict": ...} checkpoint; the positional-conv weight norm in any of its three spellings
    (torch >= 2.1 parametrizations.weight.original0/1, torch 2.0-style weight_g / weight_v as the published checkpoint was
    written under torch 2.1.2 + transformers 4.36.2, or an already materialised .weight)."""
    sd = _unwrap(sd)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    w2 = "wav2vec2."
    for i in range(7):
        p = f"{w2}feature_extractor.conv_layers.{i}"
        cw = _f32(sd[p + ".conv.weight"])
        out[f"fe{i}.w"] = cw.reshape(512, 10) if i == 0 else _conv1d_w(cw)
        out[f"fe{i}.cb"] = _f32(sd[p + ".conv.bias"])
        out[f"fe{i}.ln.g"], out[f"fe{i}.ln.b"] = _f32(sd[p + ".layer_norm.weight"]), _f32(sd[p + ".layer_norm.bias"])
    p = w2 + "feature_projection."
    out["fp.ln.g"], out["fp.ln.b"] = _f32(sd[p + "layer_norm.weight"]), _f32(sd[p + "layer_norm.bias"])
    out["fp.w"], out["fp.b"] = _f32(sd[p + "projection.weight"]), _f32(sd[p + "projection.bias"])
    pw = pos_conv_weight(sd)  # [1024, 64, 128] (out, in/groups, k)
    out["pos.w"] = _conv1d_w(pw)  # [1024, 128*64]: rows g*64..g*64+63 are group g (one grouped launch)
    out["pos.b"] = _f32(sd[w2 + "encoder.pos_conv_embed.conv.bias"])
    for l in range(12):
        p = f"{w2}encoder.layers.{l}."
        out[f"enc{l}.ln1.g"], out[f"enc{l}.ln1.b"] = _f32(sd[p + "layer_norm.weight"]), _f32(sd[p + "layer_norm.bias"])
        a = p + "attention."
        out[f"enc{l}.qkv.w"] = np.concatenate([_f32(sd[a + f"{n}_proj.weight"]) for n in "qkv"], axis=0)
        out[f"enc{l}.qkv.b"] = np.concatenate([_f32(sd[a + f"{n}_proj.bias"]) for n in "qkv"], axis=0)
        out[f"enc{l}.o.w"], out[f"enc{l}.o.b"] = _f32(sd[a + "out_proj.weight"]), _f32(sd[a + "out_proj.bias"])
        out[f"enc{l}.ln2.g"] = _f32(sd[p + "final_layer_norm.weight"])
        out[f"enc{l}.ln2.b"] = _f32(sd[p + "final_layer_norm.bias"])
        f = p + "feed_forward."
        out[f"enc{l}.ff1.w"], out[f"enc{l}.ff1.b"] = _f32(sd[f + "intermediate_dense.weight"]), _f32(sd[f + "intermediate_dense.bias"])
        out[f"enc{l}.ff2.w"], out[f"enc{l}.ff2.b"] = _f32(sd[f + "output_dense.weight"]), _f32(sd[f + "output_dense.bias"])
    out["enc.ln.g"], out["enc.ln.b"] = _f32(sd[w2 + "encoder.layer_norm.weight"]), _f32(sd[w2 + "encoder.layer_norm.bias"])
    pe = _f32(sd["tl1.positional_encoding.pe"]).reshape(-1, 1024)
    out["pe"] = np.ascontiguousarray(pe[:PE_ROWS])
    for l in (1, 2):
        t = f"tl{l}."
        a = t + "self_attention."
        out[f"tl{l}.qkv.w"] = np.concatenate(
            [_f32(sd[a + "query_w.weight"]), _f32(sd[a + "keys_w.weight"]), _f32(sd[a + "values_w.weight"])], axis=0)
        out[f"tl{l}.o.w"] = _f32(sd[a + "ff_layer_after_concat.weight"])
        out[f"tl{l}.ln1.g"] = _f32(sd[t + "add_norm_after_attention.layer_norm.weight"])
        out[f"tl{l}.ln1.b"] = _f32(sd[t + "add_norm_after_attention.layer_norm.bias"])
        out[f"tl{l}.ff1.w"], out[f"tl{l}.ff1.b"] = _f32(sd[t + "feed_forward.layer_1.weight"]), _f32(sd[t + "feed_forward.layer_1.bias"])
        out[f"tl{l}.ff2.w"], out[f"tl{l}.ff2.b"] = _f32(sd[t + "feed_forward.layer_2.weight"]), _f32(sd[t + "feed_forward.layer_2.bias"])
        out[f"tl{l}.ln2.g"] = _f32(sd[t + "add_norm_after_ff.layer_norm.weight"])
        out[f"tl{l}.ln2.b"] = _f32(sd[t + "add_norm_after_ff.layer_norm.bias"])
        # tl{l}.feed_forward.layer_norm.* exists in the state_dict but is never applied (attention_layers.py:46,50-57)
    out["td0.w"] = _conv1d_w(sd["time_downsample.0.weight"])
    out["td0.s"], out["td0.b"] = _bn_fold(sd, "time_downsample.1", AUDIO_BN_EPS, sd["time_downsample.0.bias"])
    out["td4.w"] = _conv1d_w(sd["time_downsample.4.weight"])
    out["td4.s"], out["td4.b"] = _bn_fold(sd, "time_downsample.5", AUDIO_BN_EPS, sd["time_downsample.4.bias"])
    out["fd.w"], out["fd.b"] = _f32(sd["feature_downsample.weight"]), _f32(sd["feature_downsample.bias"])
    return out


def to_blob(tensors: "OrderedDict[str, np.ndarray]") -> bytes:
    entry = struct.Struct("<96sII4qQQ")
    table_end = 16 + entry.size * len(tensors)
    off = (table_end + 63) & ~63
    recs, payload = [], []
    for name, a in tensors.items():
        a = np.ascontiguousarray(a, dtype="<f4")
        if a.ndim > 4 or len(name.encode()) > 95:
            raise ValueError(f"cannot pack {name} with shape {a.shape}")
        dims = list(a.shape) + [0] * (4 - a.ndim)
        recs.append(entry.pack(name.encode(), a.ndim, 0, *dims, off, a.nbytes))
        payload.append((off, a))
        off = (off + a.nbytes + 63) & ~63
    buf = bytearray(off)
    buf[0:8] = b"AVCERW01"
    struct.pack_into("<II", buf, 8, len(tensors), 0)
    pos = 16
    for r in recs:
        buf[pos:pos + entry.size] = r
        pos += entry.size
    for o, a in payload:
        buf[o:o + a.nbytes] = a.tobytes()
    return bytes(buf)
