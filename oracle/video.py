"""ORACLE (test infrastructure, never shipped on the product path).

CPU fp32 restatement of AVCER's visual hot path in functional torch.  No module of
/root/reference is imported here; every function cites the reference lines it restates
and is pinned against golden vectors produced by the imported reference
(tests/golden/make_golden.py -> tests/golden/*.npz).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # architectures/video.py:21,26,37,101 (eps=0.001 everywhere)
MEAN_BGR = (91.4953, 103.8827, 131.0912)  # data/utils.py:27-29
RESNET_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # video.py:105-108,165


def nearest_resize_u8(img_hwc: np.ndarray, size: int = 224) -> np.ndarray:
    """PIL Image.resize(NEAREST) (data/utils.py:34): src = floor((dst + 0.5) * in / out)."""
    h, w = img_hwc.shape[:2]
    if h == size and w == size:
        return img_hwc
    ys = np.minimum((np.arange(size) + 0.5) * (h / size), h - 1).astype(np.int64)
    xs = np.minimum((np.arange(size) + 0.5) * (w / size), w - 1).astype(np.int64)
    return img_hwc[ys][:, xs]


def pth_processing(frames_u8_hwc: np.ndarray) -> torch.Tensor:
    """data/utils.py:19-39: PILToTensor (u8 CHW) -> float32 -> flip channel axis (RGB->BGR)
    -> subtract per-channel mean.  No /255, no std.  [N,224,224,3] u8 RGB -> [N,3,224,224] f32."""
    x = torch.from_numpy(np.ascontiguousarray(frames_u8_hwc)).permute(0, 3, 1, 2).to(torch.float32)
    x = torch.flip(x, dims=(1,))
    x[:, 0] -= MEAN_BGR[0]
    x[:, 1] -= MEAN_BGR[1]
    x[:, 2] -= MEAN_BGR[2]
    return x


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _same_pad(i: int, k: int, s: int) -> int:
    """Conv2dSame.calc_same_pad, video.py:65-66 (dilation 1)."""
    return max((math.ceil(i / s) - 1) * s + (k - 1) + 1 - i, 0)


def stem(sd, x):
    """video.py:68-90,116-117: TF-'same' asymmetric pad, 7x7/2 conv, BN, ReLU, 3x3/2 max-pool (no pad)."""
    ph = _same_pad(x.shape[-2], 7, 2)
    pw = _same_pad(x.shape[-1], 7, 2)
    x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
    x = F.conv2d(x, sd["conv_layer_s2_same.weight"], None, 2)
    x = F.relu(_bn(x, sd, "batch_norm1"))
    return F.max_pool2d(x, 3, 2)


def bottleneck(sd, p, x, stride, has_ds):
    """video.py:43-60."""
    identity = x
    y = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride), sd, p + ".batch_norm1"))
    y = F.relu(_bn(F.conv2d(y, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".batch_norm2"))
    y = _bn(F.conv2d(y, sd[p + ".conv3.weight"]), sd, p + ".batch_norm3")
    if has_ds:
        identity = _bn(F.conv2d(identity, sd[p + ".i_downsample.0.weight"], None, stride), sd, p + ".i_downsample.1")
    return F.relu(y + identity)


def resnet50_forward(sd, x, taps: dict | None = None):
    """video.py:115-133.  Returns (logits [N,7], features [N,512] = fc1 output *before* ReLU, the
    tensor the forward hook of get_prob_video.py:49 captures)."""
    x = stem(sd, x)
    if taps is not None:
        taps["stem"] = x
    for li, (planes, blocks, stride) in enumerate(RESNET_STAGES, start=1):
        for b in range(blocks):
            x = bottleneck(sd, f"layer{li}.{b}", x, stride if b == 0 else 1, b == 0)
        if taps is not None:
            taps[f"layer{li}"] = x
    x = F.adaptive_avg_pool2d(x, (1, 1)).reshape(x.shape[0], -1)
    if taps is not None:
        taps["avgpool"] = x
    feats = F.linear(x, sd["fc1.weight"], sd["fc1.bias"])
    logits = F.linear(F.relu(feats), sd["fc2.weight"], sd["fc2.bias"])
    return logits, feats


def _lstm_layer(x, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTM single layer, batch_first, zero initial state, gate order i,f,g,o."""
    n, t, _ = x.shape
    hid = w_hh.shape[1]
    h = x.new_zeros(n, hid)
    c = x.new_zeros(n, hid)
    xp = F.linear(x, w_ih, b_ih)
    outs = []
    for s in range(t):
        g = xp[:, s] + F.linear(h, w_hh, b_hh)
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, dim=1)


def lstm_forward(sd, x):
    """video.py:181-185: lstm1(512->512) -> lstm2(512->256) -> fc(256->7) on the last step; raw logits."""
    x = _lstm_layer(x, sd["lstm1.weight_ih_l0"], sd["lstm1.weight_hh_l0"],
                    sd["lstm1.bias_ih_l0"], sd["lstm1.bias_hh_l0"])
    x = _lstm_layer(x, sd["lstm2.weight_ih_l0"], sd["lstm2.weight_hh_l0"],
                    sd["lstm2.bias_ih_l0"], sd["lstm2.bias_hh_l0"])
    return F.linear(x[:, -1, :], sd["fc.weight"], sd["fc.bias"])


def lstm_step(fps: float) -> int:
    """get_prob_video.py:77 (Python round = banker's rounding)."""
    return round((5 * fps) / 25)


def visual_forward(sd_static, sd_dynamic, frames_u8: np.ndarray, present: np.ndarray, fps: float,
                   batched: bool = False):
    """The per-frame loop of get_prob_video.py:91-178 for one video/clip.

    frames_u8 [T,224,224,3] RGB, present [T] bool (False = no face crop on disk for that frame).
    Returns (static_probs [T,7] float64-or-float32 rows exactly as the reference stacks them,
    dynamic_logits [T,7]) in VIDEO column order.  `batched` runs the CNN once over all present
    frames (same arithmetic per frame up to reduction order); default is the reference's batch-1 loop.
    """
    step = lstm_step(fps)
    t_total = len(present)
    zeros = np.zeros((1, 7))
    last_output = None
    window: list[np.ndarray] = []
    probs_static, probs_dynamic = [], []
    pre = None
    if batched and present.any():
        idx = np.nonzero(present)[0]
        with torch.no_grad():
            lg, ft = resnet50_forward(sd_static, pth_processing(frames_u8[idx]))
            pre = {int(i): (F.softmax(lg[j:j + 1], dim=1).numpy(), F.relu(ft[j:j + 1]).numpy())
                   for j, i in enumerate(idx)}
    for i in range(t_total):
        if present[i]:
            if pre is not None:
                output_s, feat = pre[i]
            else:
                with torch.no_grad():
                    lg, ft = resnet50_forward(sd_static, pth_processing(frames_u8[i:i + 1]))
                    output_s = F.softmax(lg, dim=1).numpy()
                    feat = F.relu(ft).numpy()
            if i % step == 0:
                window = [feat] * 10 if len(window) == 0 else window[1:] + [feat]
                lstm_in = torch.from_numpy(np.vstack(window)).unsqueeze(0)
                with torch.no_grad():
                    output_d = lstm_forward(sd_dynamic, lstm_in).numpy()
                last_output = output_d
            else:
                output_d = last_output if last_output is not None else zeros
            probs_static.append(output_s[0])
            probs_dynamic.append(output_d[0])
        else:
            window = []
            if last_output is not None:
                probs_static.append(probs_static[-1])
                probs_dynamic.append(probs_dynamic[-1])
            else:
                probs_static.append(zeros[0])
                probs_dynamic.append(zeros[0])
    return np.array(probs_static), np.array(probs_dynamic)
