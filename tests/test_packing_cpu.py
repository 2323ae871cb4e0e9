"""Real-checkpoint readiness of avcer_amd/packing.py (CPU): the spellings a published checkpoint can arrive in must pack
to the very bytes the synthetic state_dicts (reference key names, tests/golden) pack to.

  * audio: the file `epoch_N.pth` is {"epoch", "model_state_dict", ...} (audio/net_trainer/net_trainer.py:273-285, read at
    get_prob_audio_8_cl.py:58-65) and was written under torch 2.1.2 / transformers 4.36.2 (requirements.txt:42,46), whose
    weight-normed positional conv is `pos_conv_embed.conv.weight_g` / `.weight_v`; transformers 5.x names the same two
    tensors `parametrizations.weight.original0` / `original1` (what the golden generator's ExprModelV3 has).
  * detector: `Resnet50_Final.pth` carries DataParallel's "module." prefix (retina_face_predictor.py:28-33).
"""
import hashlib
from collections import OrderedDict

import numpy as np
import pytest
import torch

from avcer_amd import packing, synth

POS = "wav2vec2.encoder.pos_conv_embed.conv"


def digest(tensors):
    return hashlib.sha256(packing.to_blob(tensors)).hexdigest()


@pytest.fixture(scope="module")
def audio_sd():
    return synth.audio_state_dict(42)


@pytest.fixture(scope="module")
def audio_digest(audio_sd):
    return digest(packing.pack_audio(audio_sd))


def test_audio_checkpoint_with_weight_g_weight_v_and_trainer_wrapper(audio_sd, audio_digest):
    old = OrderedDict()
    for k, v in audio_sd.items():
        if k == POS + ".parametrizations.weight.original0":
            old[POS + ".weight_g"] = v          # same tensor, torch < 2.1 / transformers 4.36.2 name
        elif k == POS + ".parametrizations.weight.original1":
            old[POS + ".weight_v"] = v
        else:
            old[k] = v
    assert POS + ".weight_g" in old and POS + ".weight_v" in old and not any("parametrizations" in k for k in old)
    assert tuple(np.asarray(old[POS + ".weight_g"]).shape) == (1, 1, 128)          # weight_norm(dim=2): one gain per tap
    assert tuple(np.asarray(old[POS + ".weight_v"]).shape) == (1024, 64, 128)
    ckpt = {"epoch": 63, "model_state_dict": synth.to_torch(old), "optimizer_state_dict": {}, "loss": 0.0}
    assert digest(packing.pack_audio(ckpt)) == audio_digest
    # "module."-prefixed (DataParallel) variant of the same checkpoint
    ckpt["model_state_dict"] = OrderedDict(("module." + k, v) for k, v in ckpt["model_state_dict"].items())
    assert digest(packing.pack_audio(ckpt)) == audio_digest


def test_audio_checkpoint_with_materialised_pos_conv_weight(audio_sd, audio_digest):
    g = torch.from_numpy(np.asarray(audio_sd[POS + ".parametrizations.weight.original0"]))
    v = torch.from_numpy(np.asarray(audio_sd[POS + ".parametrizations.weight.original1"]))
    flat = OrderedDict((k, t) for k, t in audio_sd.items() if "parametrizations" not in k)
    flat[POS + ".weight"] = torch._weight_norm(v, g, 2)     # what remove_weight_norm / remove_parametrizations leaves behind
    assert digest(packing.pack_audio(flat)) == audio_digest
    # and the norm really is over dims (0, 1) per tap: w[:, :, k] = g[k] * v[:, :, k] / ||v[:, :, k]||
    w = packing.pos_conv_weight(audio_sd)
    k = 17
    ref = g[0, 0, k].item() * v[:, :, k].numpy() / np.linalg.norm(v[:, :, k].numpy().astype(np.float64))
    assert np.abs(w[:, :, k] - ref).max() < 1e-6 * np.abs(ref).max()


def test_seven_class_audio_checkpoint_packs_with_seven_rows():
    p = packing.pack_audio({"model_state_dict": synth.audio_state_dict(43, num_classes=7)})
    assert p["fd.w"].shape == (7, 1024) and p["fd.b"].shape == (7,)


def test_detector_checkpoint_with_module_prefix():
    sd = synth.retina_state_dict(42)
    base = digest(packing.pack_face(sd))
    pref = OrderedDict(("module." + k, v) for k, v in synth.to_torch(sd).items())
    assert digest(packing.pack_face(pref)) == base
    # the predictor strips the prefix key by key (retina_face_predictor.py:28-33): a partly prefixed dict works too
    mixed = OrderedDict((("module." + k) if i % 2 else k, v) for i, (k, v) in enumerate(sd.items()))
    assert digest(packing.pack_face(mixed)) == base


def test_static_and_dynamic_checkpoints_accept_torch_tensors_and_prefix():
    for make, pack in ((synth.static_state_dict, packing.pack_static), (synth.dynamic_state_dict, packing.pack_dynamic)):
        sd = make(42)
        base = digest(pack(sd))
        assert digest(pack(synth.to_torch(sd))) == base
        assert digest(pack(OrderedDict(("module." + k, v) for k, v in sd.items()))) == base
        halves = OrderedDict((k, torch.from_numpy(np.asarray(v)).double()) for k, v in sd.items())  # a float64 checkpoint
        assert digest(pack(halves)) == base


def test_missing_key_is_a_key_error_naming_it(audio_sd):
    broken = OrderedDict((k, v) for k, v in audio_sd.items() if k != "tl2.self_attention.keys_w.weight")
    with pytest.raises(KeyError, match="tl2.self_attention.keys_w.weight"):
        packing.pack_audio(broken)
