"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the RetinaFace-R50 detector network.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Two parts with different pinning:
  * FPN, SSH, the three heads, their wiring and the test-phase softmax follow the reference's own modules
    (retina_face/retina_face_net.py:6-101, retina_face.py:9-43,46-115) and are pinned by tests/golden/face_net.npz, which
    was produced by the reference's RetinaFace class itself.
  * The backbone is torchvision's ResNet-50 (`models.resnet50()`, requirement `torchvision>=0.3.0`,
    data/face_detection/requirements.txt:4), a third-party dependency that is neither under /root/reference nor
    installed here.  Its published definition is restated below (7x7/2 stem with padding 3, 3x3/2 max-pool with padding
    1, bottlenecks [3, 4, 6, 3] with the stride on the 3x3 convolution, BatchNorm eps 1e-5); the golden harness had to
    supply the same restatement to the reference class, so for the backbone parity is anchored on the reference's call
    site (retina_face.py:60-62, return_layers layer2/3/4) rather than on third-party code.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))
MEAN_BGR = (104.0, 117.0, 123.0)   # retina_face_predictor.py:63


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0,
                        BN_EPS)


def preprocess(frame_bgr_u8) -> torch.Tensor:
    """retina_face_predictor.py:59-65 with rgb=False: int pixels minus the mean, HWC -> 1CHW float32."""
    x = torch.from_numpy(frame_bgr_u8.astype(int) - torch.tensor(MEAN_BGR).numpy().astype(int))
    return x.permute(2, 0, 1).unsqueeze(0).float()


def backbone(sd, x):
    """torchvision ResNet-50 children conv1 .. layer4; returns the outputs of layer2, layer3, layer4."""
    x = F.relu(_bn(F.conv2d(x, sd["body.conv1.weight"], stride=2, padding=3), sd, "body.bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, (planes, blocks, stride) in enumerate(STAGES, start=1):
        for b in range(blocks):
            p = f"body.layer{li}.{b}"
            s = stride if b == 0 else 1
            y = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
            y = F.relu(_bn(F.conv2d(y, sd[p + ".conv2.weight"], stride=s, padding=1), sd, p + ".bn2"))
            y = _bn(F.conv2d(y, sd[p + ".conv3.weight"]), sd, p + ".bn3")
            if b == 0:
                x = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=s), sd, p + ".downsample.1")
            x = F.relu(y + x)
        if li >= 2:
            feats.append(x)
    return feats


def _cbr(sd, p, x, k, relu):
    y = _bn(F.conv2d(x, sd[p + ".0.weight"], padding=k // 2), sd, p + ".1")
    return F.relu(y) if relu else y          # LeakyReLU(negative_slope=0) for out_channel 256 (retina_face_net.py:47-49,80-82)


def fpn(sd, feats):
    """retina_face_net.py:76-101."""
    o1, o2, o3 = (_cbr(sd, f"fpn.output{i + 1}", f, 1, True) for i, f in enumerate(feats))
    o2 = _cbr(sd, "fpn.merge2", o2 + F.interpolate(o3, size=o2.shape[2:], mode="nearest"), 3, True)
    o1 = _cbr(sd, "fpn.merge1", o1 + F.interpolate(o2, size=o1.shape[2:], mode="nearest"), 3, True)
    return [o1, o2, o3]


def ssh(sd, p, x):
    """retina_face_net.py:42-73."""
    c3 = _cbr(sd, p + ".conv3X3", x, 3, False)
    c5_1 = _cbr(sd, p + ".conv5X5_1", x, 3, True)
    c5 = _cbr(sd, p + ".conv5X5_2", c5_1, 3, False)
    c7 = _cbr(sd, p + ".conv7x7_3", _cbr(sd, p + ".conv7X7_2", c5_1, 3, True), 3, False)
    return F.relu(torch.cat([c3, c5, c7], dim=1))


def _head(sd, p, x, per_anchor):
    y = F.conv2d(x, sd[p + ".conv1x1.weight"], sd[p + ".conv1x1.bias"])
    return y.permute(0, 2, 3, 1).contiguous().view(y.shape[0], -1, per_anchor)


def retina_forward(sd, x):
    """RetinaFace.forward in test phase (retina_face.py:95-115): (loc [1,P,4], conf [1,P,2] softmaxed, landms [1,P,10])."""
    with torch.no_grad():
        feats = [ssh(sd, f"ssh{i + 1}", f) for i, f in enumerate(fpn(sd, backbone(sd, x)))]
        loc = torch.cat([_head(sd, f"BboxHead.{i}", f, 4) for i, f in enumerate(feats)], dim=1)
        conf = torch.cat([_head(sd, f"ClassHead.{i}", f, 2) for i, f in enumerate(feats)], dim=1)
        lm = torch.cat([_head(sd, f"LandmarkHead.{i}", f, 10) for i, f in enumerate(feats)], dim=1)
    return loc, F.softmax(conf, dim=-1), lm
