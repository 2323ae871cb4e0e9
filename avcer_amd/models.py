"""Host-side mirrors of the reference's three model callables, same call signatures:

    pth_model_static(x: f32[N,3,224,224]) -> logits f32[N,7]     get_prob_video.py:22-25,103-109
        side channel activations["features"] (fc1 output, pre-ReLU)   get_prob_video.py:47-49
    pth_model_dynamic(x: f32[N,10,512])   -> logits f32[N,7]     get_prob_video.py:51-54,126-128
    audio_model(x: f32[N,T])              -> logits f32[N,8] ((8,) when N == 1)   get_prob_audio_8_cl.py:91-92

They return torch tensors on the engine's device, so `.cpu().detach().numpy()` at the reference's call sites keeps
working.  All arithmetic happens in libavcer_hip.so.

Every call goes through `Engine.guarded`: in the default MODE_F16X3 the library's range-contract counter is read once behind
the call (one 4-byte copy), and a call during which an activation left fp16's range (|x| >= 65504) is repeated in MODE_FP32 --
the mirrors return numbers wherever the reference's fp32 modules do, NaN only where the input was NaN.
"""
from __future__ import annotations

import torch

from .engine import Engine, MODE_DEFAULT


class StaticModel:
    """Drop-in for `pth_model_static` (architectures/video.py ResNet50(7) in eval mode)."""

    def __init__(self, engine: Engine, state_dict, mode: int = MODE_DEFAULT):
        self.engine, self.mode = engine, mode
        engine.load_static(state_dict)
        self.activations = {}

    def load_state_dict(self, state_dict):
        self.engine.load_static(state_dict)

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        logits, probs, feats = self.engine.guarded(self.mode, lambda m: self.engine.static_forward_nchw(x, m))
        self.activations["features"] = feats  # what the reference's fc1 forward hook stores
        self.activations["probs"] = probs
        return logits

    forward = __call__

    def extract_features(self, x: torch.Tensor) -> torch.Tensor:
        return self.engine.guarded(self.mode, lambda m: self.engine.static_forward_nchw(x, m))[2]

    def predict_frames(self, frames_u8: torch.Tensor):
        """Fused pth_processing + forward + softmax on raw u8 RGB tiles [N,H,W,3] (data/utils.py:19-39)."""
        return self.engine.guarded(self.mode, lambda m: self.engine.static_forward(frames_u8, m))


class DynamicModel:
    """Drop-in for `pth_model_dynamic` (architectures/video.py LSTMPyTorch in eval mode)."""

    def __init__(self, engine: Engine, state_dict, mode: int = MODE_DEFAULT):
        self.engine = engine
        self.mode = mode
        engine.load_dynamic(state_dict)

    def load_state_dict(self, state_dict):
        self.engine.load_dynamic(state_dict)

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self.engine.guarded(self.mode, lambda m: self.engine.dynamic_forward(x, m))

    forward = __call__


class AudioModel:
    """Drop-in for `audio_model` (architectures/audio_8_cl.py ExprModelV3 / audio_7_cl.py ExprModelV2, eval mode).
    Input is the already normalised window, exactly what the reference passes (get_prob_audio_8_cl.py:87-92)."""

    def __init__(self, engine: Engine, state_dict, mode: int = MODE_DEFAULT):
        self.engine, self.mode = engine, mode
        engine.load_audio(state_dict)

    def load_state_dict(self, state_dict):
        self.engine.load_audio(state_dict)

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def __call__(self, x: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        if x.dim() == 1:
            x = x[None]
        out = self.engine.guarded(self.mode, lambda m: self.engine.audio_forward(x, normalize=normalize, mode=m))
        return out.squeeze(0) if out.shape[0] == 1 else out  # `x.squeeze()` at audio_8_cl.py:188

    forward = __call__
