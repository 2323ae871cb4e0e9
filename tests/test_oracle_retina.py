"""The oracle's RetinaFace-R50 restatement against vectors produced by the reference's own RetinaFace class
(FPN / SSH / heads unmodified; torchvision backbone restated in the harness) -- tests/golden/make_golden.py gen_face_net."""
import numpy as np
import pytest
import torch

from avcer_amd import synth
from oracle import face as of
from oracle import retina as orf


@pytest.fixture(scope="module")
def sd_retina():
    return synth.to_torch(synth.retina_state_dict(42))


@pytest.mark.parametrize("name", ["a", "b"])
def test_network_matches_reference_class(golden, sd_retina, name):
    g = golden("face_net")
    h, w = (int(v) for v in g[f"{name}_size"])
    frame = synth.video_frames(900, 1, h, w)[0]
    loc, conf, lm = orf.retina_forward(sd_retina, orf.preprocess(frame))
    assert loc.shape[1] == len(of.prior_boxes((h, w)))            # one row per anchor of PriorBox
    np.testing.assert_allclose(loc[0].numpy(), g[f"{name}_loc"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(conf[0].numpy(), g[f"{name}_conf"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(lm[0].numpy(), g[f"{name}_landms"], rtol=0, atol=2e-5)


def test_intermediate_statistics(golden, sd_retina):
    g = golden("face_net")
    frame = synth.video_frames(900, 1, 96, 128)[0]
    with torch.no_grad():
        feats = orf.backbone(sd_retina, orf.preprocess(frame))
        pyr = orf.fpn(sd_retina, feats)
    for k, t in (("body1", feats[0]), ("body2", feats[1]), ("body3", feats[2]), ("fpn1", pyr[0]), ("fpn3", pyr[2])):
        st = np.array([t.mean().item(), t.abs().max().item(), t.std().item()])
        np.testing.assert_allclose(st, g[f"a_{k}_stats"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(t.reshape(-1)[:16].numpy(), g[f"a_{k}_head16"], rtol=0, atol=2e-5)
