"""Deterministic synthetic weights and inputs (repo-owned counter-based generator).

Real AVCER checkpoints are not in the reference repo (README.md:15 links a
Google Drive; the audio model comes from the HF hub,
src/get_prob_audio_8_cl.py:53) and there is no network.  Parity and throughput
are therefore measured on synthetic tensors that are bit-identical on every
machine: value i of tensor `name` is a pure function of (seed, name, i)
through splitmix64, independent of numpy's Generator implementation.

The dictionaries returned here use the *reference's* state_dict key names
(SURVEY.md section 8a), so the same packers accept a real checkpoint.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

_U64 = np.uint64


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def raw_u64(seed: int, name: str, n: int) -> np.ndarray:
    """n 64-bit words, a pure function of (seed, name, index)."""
    key = _fnv1a64(f"{seed}:{name}")
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([key], dtype=_U64))[0]
        idx = np.arange(n, dtype=_U64) * _U64(0xD1342543DE82EF95) + base
    return _splitmix64(idx)


def uniform01(seed: int, name: str, n: int) -> np.ndarray:
    """float32 uniform in [0, 1) with 24 random bits."""
    r = raw_u64(seed, name, n) >> _U64(40)
    return r.astype(np.float32) * np.float32(1.0 / (1 << 24))


def uniform(seed: int, name: str, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    return (u * np.float32(hi - lo) + np.float32(lo)).astype(np.float32).reshape(shape)


def centered(seed: int, name: str, shape, std: float) -> np.ndarray:
    """Zero-mean uniform with the requested standard deviation."""
    a = std * math.sqrt(3.0)
    return uniform(seed, name, shape, -a, a)


def u8(seed: int, name: str, shape) -> np.ndarray:
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.uint8)
    step = 1 << 22
    for s in range(0, n, step):
        m = min(step, n - s)
        words = raw_u64(seed, f"{name}#{s}", (m + 7) // 8)
        out[s:s + m] = words.view(np.uint8)[:m]
    return out.reshape(shape)


# --------------------------------------------------------------------------- inputs
def face_frames(seed: int, n: int, h: int = 224, w: int = 224) -> np.ndarray:
    """uint8 RGB face tiles [n, h, w, 3] (uniform [0,255])."""
    return u8(seed, "frames", (n, h, w, 3))


def waveforms(seed: int, n: int, t: int) -> np.ndarray:
    """float32 waveforms [n, t]: ~N(0, 0.1) (Irwin-Hall of 4 uniforms), clipped to [-1, 1]."""
    acc = np.zeros(n * t, dtype=np.float32)
    for k in range(4):
        acc += uniform01(seed, f"wav{k}", n * t)
    x = (acc - np.float32(2.0)) * np.float32(0.1 * math.sqrt(3.0))
    return np.clip(x, -1.0, 1.0).astype(np.float32).reshape(n, t)


# --------------------------------------------------------------------------- static CNN
_RESNET_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))


def _bn(sd, seed, prefix, c, gamma_lo=0.8, gamma_hi=1.2):
    sd[prefix + ".weight"] = uniform(seed, prefix + ".weight", (c,), gamma_lo, gamma_hi)
    sd[prefix + ".bias"] = uniform(seed, prefix + ".bias", (c,), -0.1, 0.1)
    sd[prefix + ".running_mean"] = uniform(seed, prefix + ".running_mean", (c,), -0.1, 0.1)
    sd[prefix + ".running_var"] = uniform(seed, prefix + ".running_var", (c,), 0.5, 2.0)
    sd[prefix + ".num_batches_tracked"] = np.array(0, dtype=np.int64)


def _conv2d(sd, seed, name, cout, cin, k, gain=2.0):
    std = math.sqrt(gain / (cin * k * k))
    sd[name] = centered(seed, name, (cout, cin, k, k), std)


def static_state_dict(seed: int = 42) -> "OrderedDict[str, np.ndarray]":
    """Keys of architectures/video.py:93-166 ResNet50(7).state_dict()."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    # inputs are raw pixel values minus a mean (|x| ~ 75), no /255 (data/utils.py:24-30)
    _conv2d(sd, seed, "conv_layer_s2_same.weight", 64, 3, 7, gain=2.0 / (75.0 * 75.0))
    _bn(sd, seed, "batch_norm1", 64)
    cin = 64
    for li, (planes, blocks, stride) in enumerate(_RESNET_STAGES, start=1):
        for b in range(blocks):
            p = f"layer{li}.{b}"
            _conv2d(sd, seed, p + ".conv1.weight", planes, cin, 1)
            _bn(sd, seed, p + ".batch_norm1", planes)
            _conv2d(sd, seed, p + ".conv2.weight", planes, planes, 3)
            _bn(sd, seed, p + ".batch_norm2", planes)
            _conv2d(sd, seed, p + ".conv3.weight", planes * 4, planes, 1)
            _bn(sd, seed, p + ".batch_norm3", planes * 4, 0.3, 0.5)
            if b == 0:
                _conv2d(sd, seed, p + ".i_downsample.0.weight", planes * 4, cin, 1, gain=1.0)
                _bn(sd, seed, p + ".i_downsample.1", planes * 4)
            cin = planes * 4
    sd["fc1.weight"] = centered(seed, "fc1.weight", (512, 2048), math.sqrt(4.0 / 2048))
    sd["fc1.bias"] = uniform(seed, "fc1.bias", (512,), -0.1, 0.1)
    sd["fc2.weight"] = centered(seed, "fc2.weight", (7, 512), math.sqrt(4.0 / 512))
    sd["fc2.bias"] = uniform(seed, "fc2.bias", (7,), -0.1, 0.1)
    return sd


# --------------------------------------------------------------------------- RetinaFace-R50 detector (row f4)
def retina_state_dict(seed: int = 42) -> "OrderedDict[str, np.ndarray]":
    """Keys of data/face_detection/ibug/face_detection/retina_face/retina_face.py:46-76 RetinaFace(cfg_re50).state_dict():
    `body.*` = torchvision ResNet-50 children up to layer4 (IntermediateLayerGetter keeps their names), `fpn.*`,
    `ssh{1,2,3}.*`, `ClassHead / BboxHead / LandmarkHead .{0,1,2}.conv1x1.*`."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    _conv2d(sd, seed, "body.conv1.weight", 64, 3, 7, gain=2.0 / (60.0 * 60.0))   # pixels minus (104, 117, 123)
    _bn(sd, seed, "body.bn1", 64)
    cin = 64
    for li, (planes, blocks, stride) in enumerate(_RESNET_STAGES, start=1):
        for b in range(blocks):
            p = f"body.layer{li}.{b}"
            _conv2d(sd, seed, p + ".conv1.weight", planes, cin, 1)
            _bn(sd, seed, p + ".bn1", planes)
            _conv2d(sd, seed, p + ".conv2.weight", planes, planes, 3)
            _bn(sd, seed, p + ".bn2", planes)
            _conv2d(sd, seed, p + ".conv3.weight", planes * 4, planes, 1)
            _bn(sd, seed, p + ".bn3", planes * 4, 0.3, 0.5)
            if b == 0:
                _conv2d(sd, seed, p + ".downsample.0.weight", planes * 4, cin, 1, gain=1.0)
                _bn(sd, seed, p + ".downsample.1", planes * 4)
            cin = planes * 4
    for i, c in enumerate((512, 1024, 2048), start=1):
        _conv2d(sd, seed, f"fpn.output{i}.0.weight", 256, c, 1)
        _bn(sd, seed, f"fpn.output{i}.1", 256)
    for i in (1, 2):
        _conv2d(sd, seed, f"fpn.merge{i}.0.weight", 256, 256, 3)
        _bn(sd, seed, f"fpn.merge{i}.1", 256)
    for i in (1, 2, 3):
        for name, co, ci in (("conv3X3", 128, 256), ("conv5X5_1", 64, 256), ("conv5X5_2", 64, 64), ("conv7X7_2", 64, 64),
                             ("conv7x7_3", 64, 64)):
            _conv2d(sd, seed, f"ssh{i}.{name}.0.weight", co, ci, 3)
            _bn(sd, seed, f"ssh{i}.{name}.1", co)
    for head, n in (("ClassHead", 4), ("BboxHead", 8), ("LandmarkHead", 20)):
        for i in range(3):
            _conv2d(sd, seed, f"{head}.{i}.conv1x1.weight", n, 256, 1, gain=1.0)
            sd[f"{head}.{i}.conv1x1.bias"] = uniform(seed, f"{head}.{i}.conv1x1.bias", (n,), -0.2, 0.2)
    return sd


def video_frames(seed: int, n: int, h: int, w: int) -> np.ndarray:
    """uint8 BGR video frames [n, h, w, 3] for the detector."""
    return u8(seed, "video", (n, h, w, 3))


# --------------------------------------------------------------------------- dynamic LSTM
def dynamic_state_dict(seed: int = 42) -> "OrderedDict[str, np.ndarray]":
    """Keys of architectures/video.py:169-185 LSTMPyTorch().state_dict()."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, inp, hid in (("lstm1", 512, 512), ("lstm2", 512, 256)):
        k = 1.0 / math.sqrt(hid)
        sd[f"{name}.weight_ih_l0"] = uniform(seed, f"{name}.weight_ih_l0", (4 * hid, inp), -k, k)
        sd[f"{name}.weight_hh_l0"] = uniform(seed, f"{name}.weight_hh_l0", (4 * hid, hid), -k, k)
        sd[f"{name}.bias_ih_l0"] = uniform(seed, f"{name}.bias_ih_l0", (4 * hid,), -k, k)
        sd[f"{name}.bias_hh_l0"] = uniform(seed, f"{name}.bias_hh_l0", (4 * hid,), -k, k)
    sd["fc.weight"] = centered(seed, "fc.weight", (7, 256), math.sqrt(8.0 / 256))
    sd["fc.bias"] = uniform(seed, "fc.bias", (7,), -0.2, 0.2)
    return sd


# --------------------------------------------------------------------------- audio ExprModelV3
W2V_CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
W2V_CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)
W2V_CONV_DIM = 512
W2V_HIDDEN = 1024
W2V_LAYERS = 12
W2V_HEADS = 16
W2V_FFN = 4096
W2V_POS_K = 128
W2V_POS_GROUPS = 16


def _linear(sd, seed, prefix, out_f, in_f, bias=True, gain=1.0):
    sd[prefix + ".weight"] = centered(seed, prefix + ".weight", (out_f, in_f), math.sqrt(gain / in_f))
    if bias:
        sd[prefix + ".bias"] = uniform(seed, prefix + ".bias", (out_f,), -0.05, 0.05)


def _ln(sd, seed, prefix, c):
    sd[prefix + ".weight"] = uniform(seed, prefix + ".weight", (c,), 0.8, 1.2)
    sd[prefix + ".bias"] = uniform(seed, prefix + ".bias", (c,), -0.1, 0.1)


def positional_encoding(max_len: int = 5000, d_model: int = 1024) -> np.ndarray:
    """The sinusoid buffer of architectures/attention_layers.py:194-211, computed in float32
    with torch so that it is bit-identical to the registered `pe` buffer."""
    import torch

    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, 1, d_model)
    pe[:, 0, 0::2] = torch.sin(position * div_term)
    pe[:, 0, 1::2] = torch.cos(position * div_term)
    return pe.permute(1, 0, 2).contiguous().numpy()


def audio_state_dict(seed: int = 42, num_classes: int = 8) -> "OrderedDict[str, np.ndarray]":
    """Keys of architectures/audio_8_cl.py:131-190 ExprModelV3(config).state_dict() with the
    audeering/wav2vec2-large-robust-12-ft-emotion-msp-dim config."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    w = "wav2vec2."
    sd[w + "masked_spec_embed"] = uniform(seed, w + "masked_spec_embed", (W2V_HIDDEN,), 0.0, 1.0)
    cin = 1
    for i, k in enumerate(W2V_CONV_KERNEL):
        p = f"{w}feature_extractor.conv_layers.{i}"
        sd[p + ".conv.weight"] = centered(seed, p + ".conv.weight", (W2V_CONV_DIM, cin, k), math.sqrt(2.0 / (cin * k)))
        sd[p + ".conv.bias"] = uniform(seed, p + ".conv.bias", (W2V_CONV_DIM,), -0.05, 0.05)
        _ln(sd, seed, p + ".layer_norm", W2V_CONV_DIM)
        cin = W2V_CONV_DIM
    _ln(sd, seed, w + "feature_projection.layer_norm", W2V_CONV_DIM)
    _linear(sd, seed, w + "feature_projection.projection", W2V_HIDDEN, W2V_CONV_DIM)
    pc = w + "encoder.pos_conv_embed.conv"
    sd[pc + ".bias"] = uniform(seed, pc + ".bias", (W2V_HIDDEN,), -0.05, 0.05)
    sd[pc + ".parametrizations.weight.original0"] = uniform(
        seed, pc + ".g", (1, 1, W2V_POS_K), 1.0, 3.0)
    sd[pc + ".parametrizations.weight.original1"] = centered(
        seed, pc + ".v", (W2V_HIDDEN, W2V_HIDDEN // W2V_POS_GROUPS, W2V_POS_K), 0.02)
    _ln(sd, seed, w + "encoder.layer_norm", W2V_HIDDEN)
    for i in range(W2V_LAYERS):
        p = f"{w}encoder.layers.{i}"
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            _linear(sd, seed, f"{p}.attention.{proj}", W2V_HIDDEN, W2V_HIDDEN)
        _ln(sd, seed, p + ".layer_norm", W2V_HIDDEN)
        _linear(sd, seed, p + ".feed_forward.intermediate_dense", W2V_FFN, W2V_HIDDEN, gain=2.0)
        _linear(sd, seed, p + ".feed_forward.output_dense", W2V_HIDDEN, W2V_FFN)
        _ln(sd, seed, p + ".final_layer_norm", W2V_HIDDEN)
    pe = positional_encoding()
    for tl in ("tl1", "tl2"):
        for name in ("query_w", "keys_w", "values_w", "ff_layer_after_concat"):
            _linear(sd, seed, f"{tl}.self_attention.{name}", 1024, 1024, bias=False, gain=2.0)
        _linear(sd, seed, f"{tl}.feed_forward.layer_1", 1024, 1024, gain=2.0)
        _linear(sd, seed, f"{tl}.feed_forward.layer_2", 1024, 1024)
        _ln(sd, seed, f"{tl}.feed_forward.layer_norm", 1024)  # present in the state_dict, never applied
        _ln(sd, seed, f"{tl}.add_norm_after_attention.layer_norm", 1024)
        _ln(sd, seed, f"{tl}.add_norm_after_ff.layer_norm", 1024)
        sd[f"{tl}.positional_encoding.pe"] = pe
    td = "time_downsample"
    sd[td + ".0.weight"] = centered(seed, td + ".0.weight", (1024, 1024, 5), math.sqrt(2.0 / (1024 * 5)))
    sd[td + ".0.bias"] = uniform(seed, td + ".0.bias", (1024,), -0.05, 0.05)
    _bn(sd, seed, td + ".1", 1024)
    sd[td + ".4.weight"] = centered(seed, td + ".4.weight", (1024, 1024, 3), math.sqrt(2.0 / (1024 * 3)))
    sd[td + ".4.bias"] = uniform(seed, td + ".4.bias", (1024,), -0.05, 0.05)
    _bn(sd, seed, td + ".5", 1024)
    sd["feature_downsample.weight"] = centered(
        seed, "feature_downsample.weight", (num_classes, 1024), math.sqrt(16.0 / 1024))
    sd["feature_downsample.bias"] = uniform(seed, "feature_downsample.bias", (num_classes,), -0.2, 0.2)
    return sd


def to_torch(sd):
    import torch

    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd.items())
