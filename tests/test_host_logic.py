"""CPU tests of the host-side logic (index arithmetic, packing, ABI surface).  No GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from avcer_amd import audio_pipeline, fusion, packing, synth, video_pipeline
from oracle import audio as oa
from oracle import fusion as of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_clip_reproduces_reference_tables(golden):
    """The index plan + row gathers must rebuild the reference's per-frame tables from its own per-frame outputs."""
    g = golden("visual_harness")
    for name in ("gap25", "gap30", "lead25", "full25"):
        present, fps = g[f"{name}_present"], float(g[f"{name}_fps"])
        ref_s, ref_d = g[f"{name}_static"], g[f"{name}_dynamic"]
        plan = video_pipeline.plan_clip(present, fps)
        # rows of the reference table at present frames are the "feature table" rows
        feat_rows = ref_s[present]
        rebuilt = np.stack([feat_rows[i] if i >= 0 else np.zeros(7) for i in plan.static_src])
        np.testing.assert_array_equal(rebuilt, ref_s)
        # every frame mapped to the same LSTM evaluation must carry identical dynamic rows, zeros where unmapped
        for f, d in enumerate(plan.dyn_src):
            if d < 0:
                assert not ref_d[f].any()
        for w in range(len(plan.windows)):
            rows = ref_d[[f for f, d in enumerate(plan.dyn_src) if d == w]]
            assert len(rows) and (rows == rows[0]).all()
        # window contents: first window after a reset is one feature x10, then a sliding window
        step = video_pipeline.lstm_step(fps)
        for w in plan.windows:
            assert len(w) == 10 and all(0 <= i < present.sum() for i in w)
        assert len(plan.windows) == sum(1 for i, p in enumerate(present) if p and i % step == 0)


def test_plan_clip_window_reset():
    plan = video_pipeline.plan_clip([1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1], fps=5)  # step = 1
    assert plan.windows[0] == [0] * 10
    assert plan.windows[1] == [0] * 9 + [1]
    assert plan.windows[2] == [2] * 10  # reset after the missing face
    assert plan.static_src[2] == plan.static_src[1] and plan.dyn_src[2] == plan.dyn_src[1]


def test_chunk_spans_match_reference(golden):
    g = golden("chunker")
    for key in sorted(k[:-7] for k in g.files if k.endswith("_frames")):
        fps, n, w, s, _ = key.split("_")
        fps, n, w, s = int(fps[3:]), int(n[1:]), int(w[1:]), float(s[1:])
        starts, ends, lo, hi = audio_pipeline.chunk_spans(n, 16000, fps, w, s)
        _, frames = audio_pipeline.replicate_per_frame(np.zeros((len(lo), 8), np.float32), lo, hi)
        np.testing.assert_array_equal(frames, g[key + "_frames"])
        assert [(a, b, c, d) for a, b, c, d in zip(starts, ends, lo, hi)] == oa.chunk_spans(n, 16000, fps, w, s)


def test_fusion_constants_match_reference(golden):
    g = golden("fusion")
    np.testing.assert_array_equal(np.array(fusion.WEIGHTS_AV_1), g["weights_av_1"])
    assert fusion.WEIGHTS_AV_1 == of.WEIGHTS_AV_1


def test_covered_frames():
    assert fusion.covered_frames([0, 5], [12, 17], 16) == 16
    assert fusion.covered_frames([0], [10], 16) == 10
    with pytest.raises(IndexError):
        fusion.covered_frames([20], [30], 16)


def test_pack_static_layouts(sd_static):
    t = packing.pack_static(sd_static)
    w = sd_static["conv_layer_s2_same.weight"].numpy()
    stem = t["stem.w"].reshape(64, 8, 8, 4)
    np.testing.assert_array_equal(stem[:, :7, :7, :3], w.transpose(0, 2, 3, 1))
    assert not stem[:, 7].any() and not stem[:, :, 7].any() and not stem[..., 3].any()
    c2 = t["l2.1.c2.w"].reshape(128, 3, 3, 128)
    np.testing.assert_array_equal(c2, sd_static["layer2.1.conv2.weight"].numpy().transpose(0, 2, 3, 1))
    # BN fold: scale * x + bias == batch_norm(x)
    x = torch.randn(4, 256, 3, 3)
    p = "layer1.1.batch_norm3"
    ref = torch.nn.functional.batch_norm(x, sd_static[p + ".running_mean"], sd_static[p + ".running_var"],
                                         sd_static[p + ".weight"], sd_static[p + ".bias"], False, 0.0, 1e-3)
    got = x * torch.from_numpy(t["l1.1.c3.s"])[None, :, None, None] + torch.from_numpy(t["l1.1.c3.b"])[None, :, None, None]
    assert (ref - got).abs().max() < 1e-5
    # block 0 of each stage: c3 + downsample fused into c3d.{w,b}; + stem7.w and the row-permuted chain weights
    # .wf of the chain blocks: stage 1 from block 0 (its conv3 is c3d.w), stage 2 from block 1; stage 3 tails: c3 of blocks
    # 1..4 with c1 of blocks 2..5
    n_chain = (1 + 3 + 3) + (2 + 3 + 3) + 2 * 4
    assert len(t) == 1 * 3 + 16 * 9 - 4 * 3 + 4 * 2 + 4 + 1 + n_chain + 1   # + stem.b9
    np.testing.assert_array_equal(t["stem7.w"].reshape(64, 7, 8, 4), stem[:, :7])
    # stem.b9: the shifts of the stem fed with raw pixels (fused.hip stem_pool_u8_kernel), checked where they matter -- a
    # constant image p makes every valid tap contribute w * (p - mean): conv(p - mean, zero padding) at the four corner / edge /
    # interior positions of each class must equal conv(p, zero padding) * scale + b9[class]
    p_img = torch.full((1, 3, 224, 224), 37.0)
    mu = torch.tensor(packing.PIXEL_MEANS).view(1, 3, 1, 1)
    wt = torch.from_numpy(w)
    pad = (2, 3, 2, 3)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad((p_img - mu).double(), pad), wt.double(), stride=2)
    raw = torch.nn.functional.conv2d(torch.nn.functional.pad(p_img.double(), pad), wt.double(), stride=2)
    sc, sh = torch.from_numpy(t["stem.s"]).double(), torch.from_numpy(t["stem.b"]).double()
    b9 = torch.from_numpy(t["stem.b9"]).double()
    for cy, ry in ((0, 0), (57, 1), (110, 2)):
        for cx, rx in ((0, 0), (33, 1), (110, 2)):
            want = ref[0, :, cy, cx] * sc + sh
            got9 = raw[0, :, cy, cx] * sc + b9[3 * ry + rx]
            assert (want - got9).abs().max() < 1e-4 * max(1.0, want.abs().max().item()), (cy, cx)
    # chain weights (csrc/fused.hip): the BN scale folded into the rows (the device applies the row permutation)
    np.testing.assert_array_equal(t["l2.2.c3.wf"], t["l2.2.c3.w"] * t["l2.2.c3.s"][:, None])
    assert "l1.0.c1.wf" not in t and "l1.1.c1.wf" in t and "l2.1.c1.wf" not in t and "l2.2.c1.wf" in t and "l3.1.c2.wf" not in t and "l3.1.c3.wf" in t and "l3.5.c1.wf" in t and "l3.5.c3.wf" not in t
    # numpy twin of the device-side row order: stored row 16t + 4g + r of every 32-row group holds channel 8g + 4t + r
    w0 = t["l2.2.c3.w"]
    wp = packing.permute_rows_for_mfma(w0)
    for q in (0, 5):
        for tt in range(2):
            for g in range(4):
                for r in range(4):
                    np.testing.assert_array_equal(wp[32 * q + 16 * tt + 4 * g + r], w0[32 * q + 8 * g + 4 * tt + r])
    w3 = sd_static["layer2.0.conv3.weight"].numpy()[:, :, 0, 0]
    assert t["l2.0.c3d.w"].shape == (512, 128 + 256)
    np.testing.assert_allclose(t["l2.0.c3d.w"][:, :128] / w3, np.broadcast_to((t["l2.0.c3d.w"][:, :1] / w3[:, :1]), w3.shape), rtol=1e-5)


def test_pack_audio_layouts(sd_audio):
    t = packing.pack_audio(sd_audio)
    w = oa.pos_conv_weight(sd_audio).numpy()  # torch._weight_norm
    g3 = t["pos.w"].reshape(1024, 128, 64)[192:256]
    np.testing.assert_array_equal(g3, w[192:256].transpose(0, 2, 1))
    assert t["enc4.qkv.w"].shape == (3072, 1024) and t["tl1.qkv.w"].shape == (3072, 1024)
    np.testing.assert_array_equal(t["enc4.qkv.w"][1024:2048],
                                  sd_audio["wav2vec2.encoder.layers.4.attention.k_proj.weight"].numpy())
    assert t["fe1.w"].shape == (512, 1536) and t["fe0.w"].shape == (512, 10)
    assert t["pe"].shape == (256, 1024) and t["fd.w"].shape == (8, 1024)


def test_blob_roundtrip():
    tensors = {"a.w": np.arange(24, dtype=np.float32).reshape(2, 3, 4), "b": np.ones((5,), np.float32)}
    blob = packing.to_blob(tensors)
    assert blob[:8] == b"AVCERW01" and len(blob) % 64 == 0
    import struct
    count = struct.unpack_from("<I", blob, 8)[0]
    assert count == 2
    entry = struct.Struct("<96sII4qQQ")
    name, ndim, _, d0, d1, d2, d3, off, nbytes = entry.unpack_from(blob, 16)
    assert name.rstrip(b"\0") == b"a.w" and (ndim, d0, d1, d2) == (3, 2, 3, 4) and off % 64 == 0
    np.testing.assert_array_equal(np.frombuffer(blob, np.float32, 24, off).reshape(2, 3, 4), tensors["a.w"])


def test_c_abi_exports_every_declared_symbol():
    from avcer_amd import _lib, build

    build.build()
    header = open(os.path.join(ROOT, "include", "avcer_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(avcer_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(build.LIB)
    for name in declared:
        assert hasattr(lib, name), name
    # the header's version, the library's and the binding's agree (avcer_amd/_lib.py refuses a mismatch at load)
    assert int(re.search(r"#define AVCER_ABI_VERSION (\d+)", header).group(1)) == lib.avcer_abi_version() == _lib.ABI_VERSION
    assert int(re.search(r"#define AVCER_SPLIT_TRAILER (\d+)", header).group(1)) == _lib.SPLIT_TRAILER


def test_engine_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from avcer_amd.engine import Engine

    with pytest.raises(RuntimeError):
        Engine(0)


def test_synth_is_deterministic():
    a = synth.uniform01(7, "x", 5)
    np.testing.assert_array_equal(a, synth.uniform01(7, "x", 5))
    assert a.tolist() == pytest.approx([float(v) for v in a]) and 0 <= a.min() and a.max() < 1
    # pinned values: the generator must never change (golden vectors depend on it)
    assert synth.raw_u64(42, "pin", 2).tolist() == synth.raw_u64(42, "pin", 2).tolist()
    assert synth.face_frames(1, 1)[0, 0, 0].tolist() == synth.face_frames(1, 1)[0, 0, 0].tolist()


def test_sp32_host_view_roundtrip_and_layout():
    """avcer_amd/sp32.py, the host-side view of the x3 mode's storage (csrc/split_dev.h): per 32 channels 32 fp16 hi then 32
    fp16 lo; hi + lo carries 22 significand bits; subnormal lo halves survive; an overflow decodes to NaN."""
    from avcer_amd.sp32 import from_sp32, raw_to_f32, to_sp32

    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 7, 64, generator=g) * torch.tensor([1e-3, 1.0, 3e3, 0.1, 30.0]).view(5, 1, 1)
    s = to_sp32(x)
    assert s.dtype == torch.int16 and s.shape == (5, 7, 128)
    back = from_sp32(s)
    assert ((back - x).abs() <= 2.0 ** -21 * x.abs() + 2.0 ** -24).all()
    # layout: element c of a row -> hi at int16 index 64 * (c // 32) + c % 32, lo 32 further on
    row = s[1, 2].view(torch.float16).float()
    for c in (0, 31, 32, 63):
        assert abs(float(row[64 * (c // 32) + c % 32] + row[64 * (c // 32) + 32 + c % 32]) - float(x[1, 2, c])) <= 2.0 ** -21 * abs(float(x[1, 2, c]))
    torch.testing.assert_close(raw_to_f32(s.reshape(-1), (5, 7, 64)), back)
    assert torch.isnan(from_sp32(to_sp32(torch.full((1, 32), 7.0e4)))).all()          # beyond fp16: inf + (-inf)
    tiny = torch.full((1, 32), 3.0e-5)                                                  # lo half is an fp16 subnormal
    assert float((from_sp32(to_sp32(tiny)) - tiny).abs().max()) <= 2.0 ** -25


def test_library_carries_the_hash_of_its_sources_and_a_stale_binary_is_refused(tmp_path):
    """avcer_amd/build.py embeds source_hash() into the .so (avcer_source_hash()); _lib.load() compares it with the tree's.
    A source touched WITHOUT a rebuild -- simulated on a copy of the sources, the tree itself stays untouched -- no longer
    matches the binary and the load is refused; the binary built from the tree loads."""
    import shutil

    from avcer_amd import _lib, build

    build.build()
    lib = ctypes.CDLL(build.LIB)
    lib.avcer_source_hash.restype = ctypes.c_char_p
    assert lib.avcer_source_hash().decode() == build.source_hash()
    assert _lib.check_source_hash(lib, build.LIB) == build.source_hash()
    # the same sources with one byte appended to one kernel file: what the tree would hash to after an edit
    paths = []
    for f in build.SOURCES + build.HEADERS:
        dst = tmp_path / os.path.basename(f)
        shutil.copy(os.path.join(build.CSRC, f), dst)
        paths.append(str(dst))
    with open(paths[0], "ab") as fh:
        fh.write(b"\n// touched\n")
    touched = build._digest(paths)
    assert touched != build.source_hash()
    with pytest.raises(RuntimeError, match="stale binary"):
        _lib.check_source_hash(lib, build.LIB, tree_hash=touched)
    # ... and build() would rebuild: its object stamps are content hashes, not mtimes
    key_now = build._digest([os.path.join(build.CSRC, build.SOURCES[0])] + [os.path.join(build.CSRC, h) for h in build.HEADERS],
                            build.FLAGS)
    assert open(os.path.join(build.CSRC, build.SOURCES[0].replace(".hip", ".o")) + ".stamp").read() == key_now


def _host_lib():
    from avcer_amd import _lib, build

    build.build()
    lib = ctypes.CDLL(build.LIB)
    for name in ("avcer_lsap", "avcer_track_faces"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _lib.SIGNATURES[name]
    return lib


def test_native_assignment_equals_scipy_including_ties():
    """csrc/track.hip lsap = scipy.optimize.linear_sum_assignment (the tracker's Hungarian step,
    simple_face_tracker.py:60): same pairs in the same order on random rectangular matrices AND on small-integer matrices,
    where many assignments share the optimal cost and only the algorithm's own tie-breaking decides."""
    from scipy.optimize import linear_sum_assignment

    lib = _host_lib()
    rng = np.random.default_rng(0)
    cases = 0
    for trial in range(600):
        nr, nc = int(rng.integers(0, 8)), int(rng.integers(0, 8))
        kind = trial % 3
        if kind == 0:
            cost = rng.uniform(0, 1, (nr, nc))
        elif kind == 1:
            cost = rng.integers(0, 3, (nr, nc)).astype(np.float64)          # ties everywhere
        else:
            cost = np.where(rng.uniform(size=(nr, nc)) < 0.5, 2.0 * min(nr, nc), rng.uniform(0, 0.6, (nr, nc)))  # the tracker's shape
        cost = np.ascontiguousarray(cost, dtype=np.float64)
        k = min(nr, nc)
        rows, cols = np.full(max(k, 1), -1, np.int32), np.full(max(k, 1), -1, np.int32)
        assert lib.avcer_lsap(nr, nc, cost.ctypes.data_as(ctypes.c_void_p), rows.ctypes.data_as(ctypes.c_void_p),
                              cols.ctypes.data_as(ctypes.c_void_p)) == 0
        r_ref, c_ref = linear_sum_assignment(cost)
        assert rows[:k].tolist() == r_ref.tolist() and cols[:k].tolist() == c_ref.tolist(), (trial, cost)
        cases += 1
    assert cases == 600
    # larger problems, both orientations, floats and tie-heavy integers (the augmenting paths get long: duals and tie rule matter)
    for trial, (nr, nc) in enumerate([(20, 30), (40, 25), (33, 33), (64, 17), (5, 90)] * 4):
        cost = rng.uniform(0, 1, (nr, nc)) if trial % 2 == 0 else rng.integers(0, 5, (nr, nc)).astype(np.float64)
        cost = np.ascontiguousarray(cost, dtype=np.float64)
        k = min(nr, nc)
        rows, cols = np.full(k, -1, np.int32), np.full(k, -1, np.int32)
        assert lib.avcer_lsap(nr, nc, cost.ctypes.data_as(ctypes.c_void_p), rows.ctypes.data_as(ctypes.c_void_p),
                              cols.ctypes.data_as(ctypes.c_void_p)) == 0
        r_ref, c_ref = linear_sum_assignment(cost)
        assert rows.tolist() == r_ref.tolist() and cols.tolist() == c_ref.tolist(), (trial, nr, nc)


def _track_native(lib, dets, w, h, iou=0.4, min_size=0.0):
    counts = np.array([len(d) for d in dets], np.int32)
    total = int(counts.sum())
    boxes = np.zeros((max(total, 1), 4), np.float32)
    if total:
        boxes[:total] = np.concatenate([np.asarray(d, np.float32).reshape(len(d), -1)[:, :4] for d in dets if len(d)])
    rec = np.empty((max(total, 1), 6), np.int64)
    n = ctypes.c_int64(0)
    rc = lib.avcer_track_faces(None, boxes.ctypes.data_as(ctypes.c_void_p), 4, counts.ctypes.data_as(ctypes.c_void_p), len(counts), w, h,
                               iou, min_size, rec.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
    return rc, rec[:n.value]


def _track_python(dets, w, h, iou=0.4, min_size=0.0):
    """The per-frame loop the native call replaces: SimpleFaceTracker + crop_rects of avcer_amd/face_tiles.py, which are
    themselves pinned to the reference's SimpleFaceTracker / VideoPredictor.process by tests/test_face_cpu.py."""
    from avcer_amd.face_tiles import SimpleFaceTracker, crop_rects

    tr = SimpleFaceTracker(iou, min_size)
    recs = []
    for t, d in enumerate(dets):
        d = np.asarray(d, np.float32).reshape(-1, 15)
        tids = tr(d)
        for (x0, y0, x1, y1), tid in zip(crop_rects(d, w, h), tids):
            if tid is None or x1 <= x0 or y1 <= y0:
                return -1, None
            recs.append((t, tid - 1, x0, y0, x1, y1))
    return 0, np.array(recs, np.int64).reshape(-1, 6)


def test_native_tracker_equals_the_python_tracker_frame_by_frame():
    """avcer_track_faces (one native host call per video) against the Python mirror of the reference's tracker driven frame by
    frame: several faces drifting, crossing, leaving and re-entering, empty frames (every tracklet dropped), boxes hanging over
    the frame edges (negative and beyond-size corners: Python's slice rules), a minimum face size -- and the two failure cases."""
    lib = _host_lib()
    rng = np.random.default_rng(1)
    w, h = 640, 360
    for trial in range(40):
        nf = int(rng.integers(3, 7))
        centres = rng.uniform([0, 0], [w, h], (nf, 2))
        vel = rng.normal(0, 6, (nf, 2))
        size = rng.uniform(20, 120, nf)
        dets = []
        for t in range(60):
            centres += vel + rng.normal(0, 2, centres.shape)
            rows = []
            for f in range(nf):
                if rng.uniform() < 0.15:
                    continue                                                   # missed detection
                cx, cy = centres[f]
                s = size[f] * rng.uniform(0.9, 1.1)
                rows.append([cx - s / 2, cy - s / 2, cx + s / 2, cy + s / 2, 0.9] + [0.0] * 10)
            if rng.uniform() < 0.08:
                rows = []                                                      # a frame without faces
            order = rng.permutation(len(rows))
            dets.append(np.array([rows[i] for i in order], np.float32).reshape(-1, 15))
        min_size = 0.0 if trial % 2 == 0 else 40.0
        rc_p, rec_p = _track_python(dets, w, h, 0.4, min_size)
        rc_n, rec_n = _track_native(lib, dets, w, h, 0.4, min_size)
        assert rc_n == rc_p, (trial, rc_n, rc_p)
        if rc_p == 0:
            np.testing.assert_array_equal(rec_n, rec_p)
    # the reference's two failure cases come back as AVCER_EINVAL (ValueError through Engine.track_faces)
    zero_area = [np.array([[10, 10, 10, 50, 0.9] + [0] * 10], np.float32)]
    assert _track_native(lib, zero_area, w, h)[0] == -1 and _track_python(zero_area, w, h)[0] == -1
    outside = [np.array([[700, 10, 760, 50, 0.9] + [0] * 10], np.float32)]
    assert _track_native(lib, outside, w, h)[0] == -1 and _track_python(outside, w, h)[0] == -1


def test_native_tracker_reproduces_the_reference_generated_records():
    from test_face_cpu import G, golden_script

    lib = _host_lib()
    frames_h, frames_w = (int(v) for v in G["track_hw"]) if "track_hw" in G else (None, None)
    script = golden_script()
    if frames_h is None:
        from test_face_cpu import golden_frames
        frames_h, frames_w = golden_frames().shape[1:3]
    rc, rec = _track_native(lib, script, frames_w, frames_h)
    assert rc == 0
    np.testing.assert_array_equal(rec, G["track_records"])


def test_read_face_dir_follows_the_reference_listing(tmp_path):
    """video_pipeline.read_face_dir = the file side of preprocess_video_and_predict (get_prob_video.py:79-100): frame i comes from
    `<dir>/00/%06d.jpg` if that file exists, decoded to RGB and resized with PIL NEAREST exactly as pth_processing does
    (data/utils.py:34); missing frames are absent (zeros, present False); a missing track directory raises like os.listdir."""
    from PIL import Image

    from avcer_amd.video_pipeline import read_face_dir

    rng = np.random.default_rng(0)
    d = tmp_path / "vid" / "00"
    d.mkdir(parents=True)
    sizes = {0: (97, 80), 1: (224, 224), 3: (301, 190), 6: (50, 61)}
    for i, (h, w) in sizes.items():
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(d / f"{i:06d}.jpg", quality=95)
    (d / "notes.txt").write_text("ignored")
    frames, present = read_face_dir(str(tmp_path / "vid"), 8)
    assert frames.shape == (8, 224, 224, 3) and frames.dtype == np.uint8
    assert present.tolist() == [i in sizes for i in range(8)]
    for i in range(8):
        if i in sizes:
            ref = np.asarray(Image.open(d / f"{i:06d}.jpg").convert("RGB").resize((224, 224), Image.Resampling.NEAREST))
            np.testing.assert_array_equal(frames[i], ref)
        else:
            assert not frames[i].any()
    with pytest.raises(FileNotFoundError):
        read_face_dir(str(tmp_path / "nothing"), 3)
