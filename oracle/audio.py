"""ORACLE (test infrastructure, never shipped on the product path).

CPU fp32 restatement of AVCER's audio hot path.  The wav2vec2 arithmetic lives in a third-party
dependency that is NOT under /root/reference: transformers==4.36.2 (src/requirements.txt:46),
classes Wav2Vec2Model / Wav2Vec2FeatureExtractor, configured as
audeering/wav2vec2-large-robust-12-ft-emotion-msp-dim (src/get_prob_audio_8_cl.py:53-57).  Its
published algorithm for that config (feat_extract_norm="layer", do_stable_layer_norm=True) is
restated below and pinned against golden vectors produced by running the reference's own
ExprModelV3 on the installed transformers (tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)
HIDDEN = 1024
LAYERS = 12
HEADS = 16
POS_K = 128
POS_GROUPS = 16
LN_EPS = 1e-5
BN_EPS = 1e-5  # torch.nn.BatchNorm1d default, architectures/audio_8_cl.py:150,154


# ----------------------------------------------------------------------------- chunking / padding
def pad_wav(wav: torch.Tensor, max_length: int) -> torch.Tensor:
    """data/utils.py:63-71 ('repeat' padding)."""
    n = len(wav)
    if n < max_length:
        reps = (max_length + n - 1) // n
        wav = torch.cat([wav] * reps, dim=0)[:max_length]
    elif n > max_length:
        wav = wav[:max_length]
    return wav


def pad_wav_zeros(wav: torch.Tensor, max_length: int, mode: str = "constant") -> torch.Tensor:
    """data/utils.py:74-89.  'mean' pads with the chunk mean (NaN for an empty chunk), else zeros."""
    pad = max(0, max_length - wav.shape[0])
    if mode == "mean":
        return F.pad(wav, (0, pad), mode="constant", value=torch.mean(wav))
    return F.pad(wav, (0, pad), mode=mode)


def normalize(x: np.ndarray) -> np.ndarray:
    """HF Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm without attention mask
    (call site get_prob_audio_8_cl.py:88-89): (x - mean) / sqrt(var + 1e-7), numpy fp32, population var."""
    x = np.asarray(x, dtype=np.float32)
    return (x - x.mean(axis=-1, keepdims=True)) / np.sqrt(x.var(axis=-1, keepdims=True) + 1e-7)


def chunk_spans(n_samples: int, sr: int, fps: float, window: float, step: float):
    """Window starts/ends and per-window frame spans, get_prob_audio_8_cl.py:70-99.
    Returns list of (start, end, frame_lo, frame_hi) with frames range(frame_lo, frame_hi)."""
    window_a = int(window * sr)
    step_a = int(step * sr)
    out = []
    for start in range(0, n_samples + 1, step_a):
        end = min(start + window_a, n_samples)
        out.append((start, end, round(start / sr * fps), round(end / sr * fps + 1)))
    return out


def make_chunks(wav: torch.Tensor, sr: int, fps: float, window: float, step: float, padding: str):
    """get_prob_audio_8_cl.py:78-90: slice, pad, normalise.  Returns (chunks [n, window*sr] f32, spans)."""
    spans = chunk_spans(len(wav), sr, fps, window, step)
    window_a = int(window * sr)
    rows = []
    for (s, e, _, _) in spans:
        c = wav[s:e]
        c = pad_wav(c, window_a) if padding == "repeat" else pad_wav_zeros(c, window_a, mode=padding)
        rows.append(normalize(c.unsqueeze(0).numpy())[0])
    return np.stack(rows).astype(np.float32), spans


def replicate_per_frame(logits: np.ndarray, spans):
    """get_prob_audio_8_cl.py:94-101: each window's logits repeated for every frame index in its span.
    Returns (rows [sum, C], frame_idx [sum] int)."""
    rows, frames = [], []
    for lg, (_, _, lo, hi) in zip(logits, spans):
        for f in range(lo, hi):
            rows.append(lg)
            frames.append(f)
    if not rows:
        return np.zeros((0, logits.shape[1]), logits.dtype), np.zeros((0,), np.int64)
    return np.stack(rows), np.asarray(frames, dtype=np.int64)


# ----------------------------------------------------------------------------- wav2vec2 (third-party)
def feature_extractor(sd, x, taps=None):
    """Wav2Vec2FeatureEncoder with Wav2Vec2LayerNormConvLayer x7: Conv1d(+bias) -> LN over channels -> GELU(erf).
    x [B,T] -> [B,S,512] (time-major)."""
    h = x[:, None, :]
    for i, (k, s) in enumerate(zip(CONV_KERNEL, CONV_STRIDE)):
        p = f"wav2vec2.feature_extractor.conv_layers.{i}"
        h = F.conv1d(h, sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=s)
        h = h.transpose(-2, -1)
        h = F.layer_norm(h, (h.shape[-1],), sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], LN_EPS)
        h = h.transpose(-2, -1)
        h = F.gelu(h)
        if taps is not None and i == 0:
            taps["conv0"] = h
    return h.transpose(1, 2)


def pos_conv_weight(sd):
    """torch weight_norm(dim=2) of encoder.pos_conv_embed.conv: w[:,:,k] = g[k] * v[:,:,k] / ||v[:,:,k]||_F."""
    p = "wav2vec2.encoder.pos_conv_embed.conv"
    if p + ".parametrizations.weight.original0" in sd:
        g, v = sd[p + ".parametrizations.weight.original0"], sd[p + ".parametrizations.weight.original1"]
    else:  # checkpoints saved by transformers 4.36.2 / torch 2.1.2 use the old weight_norm names
        g, v = sd[p + ".weight_g"], sd[p + ".weight_v"]
    return torch._weight_norm(v, g, 2)


def encoder(sd, h, taps=None):
    """Wav2Vec2EncoderStableLayerNorm (eval, no attention mask): pos-conv add, 12 pre-LN layers, final LN."""
    w = "wav2vec2.encoder."
    pc = F.conv1d(h.transpose(1, 2), pos_conv_weight(sd), sd[w + "pos_conv_embed.conv.bias"],
                  padding=POS_K // 2, groups=POS_GROUPS)
    pc = pc[:, :, :-1]  # Wav2Vec2SamePadLayer: even kernel -> drop the last frame
    h = h + F.gelu(pc).transpose(1, 2)
    if taps is not None:
        taps["posconv"] = h
    b, s, _ = h.shape
    d = HIDDEN // HEADS
    for i in range(LAYERS):
        p = f"{w}layers.{i}"
        res = h
        x = F.layer_norm(h, (HIDDEN,), sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], LN_EPS)
        q = F.linear(x, sd[p + ".attention.q_proj.weight"], sd[p + ".attention.q_proj.bias"]) * d ** -0.5
        k = F.linear(x, sd[p + ".attention.k_proj.weight"], sd[p + ".attention.k_proj.bias"])
        v = F.linear(x, sd[p + ".attention.v_proj.weight"], sd[p + ".attention.v_proj.bias"])
        q = q.view(b, s, HEADS, d).transpose(1, 2)
        k = k.view(b, s, HEADS, d).transpose(1, 2)
        v = v.view(b, s, HEADS, d).transpose(1, 2)
        a = F.softmax(torch.matmul(q, k.transpose(-2, -1)), dim=-1)
        o = torch.matmul(a, v).transpose(1, 2).reshape(b, s, HIDDEN)
        o = F.linear(o, sd[p + ".attention.out_proj.weight"], sd[p + ".attention.out_proj.bias"])
        h = res + o
        x = F.layer_norm(h, (HIDDEN,), sd[p + ".final_layer_norm.weight"], sd[p + ".final_layer_norm.bias"], LN_EPS)
        x = F.gelu(F.linear(x, sd[p + ".feed_forward.intermediate_dense.weight"],
                            sd[p + ".feed_forward.intermediate_dense.bias"]))
        x = F.linear(x, sd[p + ".feed_forward.output_dense.weight"], sd[p + ".feed_forward.output_dense.bias"])
        h = h + x
        if taps is not None and i in (0, 5, 11):
            taps[f"layer{i}"] = h
    return F.layer_norm(h, (HIDDEN,), sd[w + "layer_norm.weight"], sd[w + "layer_norm.bias"], LN_EPS)


def wav2vec2_forward(sd, x, taps=None):
    """Wav2Vec2Model.forward(...)[0] in eval mode: [B,T] -> [B,S,1024]."""
    f = feature_extractor(sd, x, taps)
    if taps is not None:
        taps["extract"] = f
    p = "wav2vec2.feature_projection."
    f = F.layer_norm(f, (f.shape[-1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], LN_EPS)
    h = F.linear(f, sd[p + "projection.weight"], sd[p + "projection.bias"])
    if taps is not None:
        taps["proj"] = h
    h = encoder(sd, h, taps)
    if taps is not None:
        taps["w2v"] = h
    return h


# ----------------------------------------------------------------------------- first-party head
def transformer_layer(sd, tl: str, x, num_heads: int):
    """architectures/attention_layers.py:249-267 with key=value=query=x: the same PE is added to each
    (all three become x+PE, which is also the residual); bias-free Q/K/V/O (:91-96); softmax(QK^T/sqrt(d))V
    (:10-38); LN(res+attn) (:60-77); Linear->ReLU->Linear (:41-57; feed_forward.layer_norm is never applied);
    LN(res+ffn)."""
    b, s, e = x.shape
    d = e // num_heads
    xp = x + sd[tl + ".positional_encoding.pe"][:, :s]
    q = F.linear(xp, sd[tl + ".self_attention.query_w.weight"]).view(b, s, num_heads, d).transpose(1, 2)
    k = F.linear(xp, sd[tl + ".self_attention.keys_w.weight"]).view(b, s, num_heads, d).transpose(1, 2)
    v = F.linear(xp, sd[tl + ".self_attention.values_w.weight"]).view(b, s, num_heads, d).transpose(1, 2)
    a = F.softmax(torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(d), dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).contiguous().view(b, s, e)
    o = F.linear(o, sd[tl + ".self_attention.ff_layer_after_concat.weight"])
    p = tl + ".add_norm_after_attention.layer_norm"
    y = F.layer_norm(o + xp, (e,), sd[p + ".weight"], sd[p + ".bias"], LN_EPS)
    f = F.linear(F.relu(F.linear(y, sd[tl + ".feed_forward.layer_1.weight"], sd[tl + ".feed_forward.layer_1.bias"])),
                 sd[tl + ".feed_forward.layer_2.weight"], sd[tl + ".feed_forward.layer_2.bias"])
    p = tl + ".add_norm_after_ff.layer_norm"
    return F.layer_norm(f + y, (e,), sd[p + ".weight"], sd[p + ".bias"], LN_EPS)


def head(sd, x):
    """architectures/audio_8_cl.py:146-159,185-190: permute, Conv1d k5 s3 dil2, BN, MaxPool(5), ReLU, Conv1d k3, BN,
    global avg, ReLU, squeeze, Linear -> raw logits.  Returns [B,C] ((C,) when B == 1, as `.squeeze()` does)."""
    t = "time_downsample"
    h = x.permute(0, 2, 1)
    h = F.conv1d(h, sd[t + ".0.weight"], sd[t + ".0.bias"], stride=3, dilation=2)
    h = F.batch_norm(h, sd[t + ".1.running_mean"], sd[t + ".1.running_var"], sd[t + ".1.weight"], sd[t + ".1.bias"],
                     False, 0.0, BN_EPS)
    h = F.relu(F.max_pool1d(h, 5))
    h = F.conv1d(h, sd[t + ".4.weight"], sd[t + ".4.bias"])
    h = F.batch_norm(h, sd[t + ".5.running_mean"], sd[t + ".5.running_var"], sd[t + ".5.weight"], sd[t + ".5.bias"],
                     False, 0.0, BN_EPS)
    h = F.relu(F.adaptive_avg_pool1d(h, 1))
    h = h.squeeze()
    return F.linear(h, sd["feature_downsample.weight"], sd["feature_downsample.bias"])


def expr_model_v3_forward(sd, x, taps=None):
    """architectures/audio_8_cl.py:179-190."""
    h = wav2vec2_forward(sd, x, taps)
    h = transformer_layer(sd, "tl1", h, 32)
    if taps is not None:
        taps["tl1"] = h
    h = transformer_layer(sd, "tl2", h, 16)
    if taps is not None:
        taps["tl2"] = h
    return head(sd, h)


def audio_forward(sd, wav: torch.Tensor, sr: int, fps: float, window: float = 4, step: float = 0.5,
                  padding: str = "mean", batched: bool = True):
    """EmotionRecognition.load_audio_features, get_prob_audio_8_cl.py:68-126, without file I/O.
    Returns (per-frame logits rows [sum, C], frame_idx [sum])."""
    chunks, spans = make_chunks(wav, sr, fps, window, step, padding)
    with torch.no_grad():
        if batched:
            lg = expr_model_v3_forward(sd, torch.from_numpy(chunks))
            lg = lg.reshape(len(chunks), -1).numpy()
        else:
            lg = np.stack([expr_model_v3_forward(sd, torch.from_numpy(c[None])).numpy() for c in chunks])
    return replicate_per_frame(lg, spans)
