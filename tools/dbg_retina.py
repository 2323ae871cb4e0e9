import sys, numpy as np, torch
sys.path.insert(0, '.')
from avcer_amd import synth
from avcer_amd.engine import Engine
from oracle import retina as orf
eng = Engine(0)
sd = synth.to_torch(synth.retina_state_dict(42))
eng.load_face(sd)
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (96, 128)
frame = synth.video_frames(900, 1, H, W)
x = orf.preprocess(frame[0])
with torch.no_grad():
    feats = orf.backbone(sd, x)
    pyr = orf.fpn(sd, feats)
    s1 = orf.ssh(sd, "ssh1", pyr[0])
    import torch.nn.functional as F
    lat = [orf._cbr(sd, f"fpn.output{i + 1}", f, 1, True) for i, f in enumerate(feats)]
    sum2 = lat[1] + F.interpolate(lat[2], size=lat[1].shape[2:], mode="nearest")
with torch.no_grad():
    import torch.nn.functional as F2
    st = F2.max_pool2d(F2.relu(orf._bn(F2.conv2d(x, sd["body.conv1.weight"], stride=2, padding=3), sd, "body.bn1")), 3, 2, 1)
    import torch.nn.functional as F3
    p0 = "body.layer1.0"
    c1 = F3.relu(orf._bn(F3.conv2d(st, sd[p0 + ".conv1.weight"]), sd, p0 + ".bn1"))
    c2 = F3.relu(orf._bn(F3.conv2d(c1, sd[p0 + ".conv2.weight"], padding=1), sd, p0 + ".bn2"))
    c3 = orf._bn(F3.conv2d(c2, sd[p0 + ".conv3.weight"]), sd, p0 + ".bn3")
    b0 = F3.relu(c3 + orf._bn(F3.conv2d(st, sd[p0 + ".downsample.0.weight"]), sd, p0 + ".downsample.1"))
    blk = {}
    xx = st
    for li, (planes, blocks, stride) in enumerate(orf.STAGES, start=1):
        for b in range(blocks):
            pp = f"body.layer{li}.{b}"
            ss = stride if b == 0 else 1
            y = F3.relu(orf._bn(F3.conv2d(xx, sd[pp + ".conv1.weight"]), sd, pp + ".bn1"))
            y = F3.relu(orf._bn(F3.conv2d(y, sd[pp + ".conv2.weight"], stride=ss, padding=1), sd, pp + ".bn2"))
            y = orf._bn(F3.conv2d(y, sd[pp + ".conv3.weight"]), sd, pp + ".bn3")
            if b == 0:
                xx = orf._bn(F3.conv2d(xx, sd[pp + ".downsample.0.weight"], stride=ss), sd, pp + ".downsample.1")
            xx = F3.relu(y + xx)
            if li <= 2:
                blk[f"face_blk{li}_{b}"] = xx
refs = {**blk, "face_l1b0_c1": c1, "face_l1b0_c2": c2, "face_l1b0": b0, "face_pool": st, "face_lat1": lat[0], "face_lat2": lat[1], "face_lat3": lat[2], "face_sum2": sum2, "face_fpn2": pyr[1],"face_body1": feats[0], "face_body2": feats[1], "face_body3": feats[2], "face_fpn1": pyr[0], "face_ssh1": s1}
for name, ref in refs.items():
    r = ref[0].permute(1, 2, 0).contiguous().reshape(-1).numpy()   # NHWC
    dst = eng.debug_tap(name, r.size)
    eng.face_forward(frame, 0)
    torch.cuda.synchronize()
    g = dst.cpu().numpy()
    print(name, "copied", eng.debug_tap_copied(), "max|d|", np.abs(g - r).max(), "ref absmax", np.abs(r).max())
