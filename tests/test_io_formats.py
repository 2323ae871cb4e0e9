"""Row f2: CSV wire formats and the dataset-level fusion driver, pinned by files/outputs of the reference
(tests/golden/make_golden.py gen_dataset_fusion runs get_pred_av.get_c_expr_db_pred on them)."""
import os

import numpy as np
import pytest

from avcer_amd import io_formats as iof

VIDEOS = ("vidA", "vidB")


def _materialise(g, tmp_path):
    root = tmp_path / "preds"
    (root / "video").mkdir(parents=True)
    (root / "audio" / "modelA").mkdir(parents=True)
    for v in VIDEOS:
        for kind in ("static", "dynamic"):
            (root / "video" / f"{kind}__{v}.csv").write_bytes(g[f"{v}_{kind}_csv"].tobytes())
        (root / "audio" / "modelA" / f"{v}.csv").write_bytes(g[f"{v}_audio_csv"].tobytes())
    fmt = tmp_path / "prediction_file_format.csv"
    fmt.write_text("image_location\n" + "\n".join(g["format_rows"].tolist()) + "\n")
    return str(root), str(fmt)


def test_writers_are_byte_compatible_with_the_reference(golden, tmp_path):
    g = golden("dataset_fusion")
    for v in VIDEOS:
        dyn_path, stat_path = iof.write_visual_csvs(g[f"{v}_stat"], g[f"{v}_dyn"], str(tmp_path / "w"), v)
        assert open(stat_path, "rb").read() == g[f"{v}_static_csv"].tobytes()
        assert open(dyn_path, "rb").read() == g[f"{v}_dynamic_csv"].tobytes()
        p = iof.write_audio_csv(g[f"{v}_aud_rows"], g[f"{v}_aud_frames"], str(tmp_path / "w"), "modelA", v)
        assert open(p, "rb").read() == g[f"{v}_audio_csv"].tobytes()
    assert iof.write_audio_csv(np.zeros((1, 8), np.float32), [0], str(tmp_path / "w"), "m", "6-30-1920x1080").endswith(
        "6-30-1920x1080_right.csv")


def test_readers_and_alignment(golden, tmp_path):
    g = golden("dataset_fusion")
    root, fmt = _materialise(g, tmp_path)
    rows, frames = iof.read_audio_csv(os.path.join(root, "audio", "modelA", "vidB.csv"))
    assert not np.isnan(rows).any() and len(rows) == len(g["vidB_aud_rows"]) - 1  # the NaN tail row is dropped
    np.testing.assert_allclose(iof.read_visual_csv(os.path.join(root, "video", "static__vidB.csv")), g["vidB_stat"], rtol=1e-6)
    listed = [r for r in g["format_rows"].tolist() if r.startswith("vidB/")]
    sel, aud_sel, aud_pos = iof.align_video(33, frames, listed, "vidB")
    assert [f"vidB/{f + 1:05d}.jpg" for f in sel] == listed
    assert aud_sel.max() == 26 and len(aud_pos) == len(sel) and aud_pos.max() == len(aud_sel) - 1
    assert (np.diff(aud_pos) >= 0).all() and aud_pos[-1] == aud_pos[-2]  # tail repeats the last audio row


@pytest.mark.gpu
def test_dataset_fusion_reproduces_the_reference_submission(engine, golden, tmp_path):
    g = golden("dataset_fusion")
    root, fmt = _materialise(g, tmp_path)
    locs, pred, txt = iof.dataset_fusion(engine, fmt, root, ["video", "audio", "modelA"], list(VIDEOS),
                                         save_path=str(tmp_path / "out"))
    ref = g["submission_txt"].tobytes().decode()
    got = open(txt).read()
    assert os.path.basename(txt) == "C_EXPR_DB_av_sd_w_False_True.txt"
    assert got.splitlines()[0] == ref.splitlines()[0]
    assert [l.split(",")[0] for l in got.splitlines()] == [l.split(",")[0] for l in ref.splitlines()]
    assert got == ref


# ----------------------------------------------------------------------------- row f3: 7-class ("Acl7") and video-only fusion
VIDEOS7 = ("vidC", "vidD")
MODEL7 = "7cl-FLW-ExprModelV2-2024.03.04-11.52.11"  # get_pred_av.py:364


def _materialise7(g, tmp_path):
    root = tmp_path / "preds"
    (root / "video").mkdir(parents=True)
    (root / "audio_repeat_1" / MODEL7).mkdir(parents=True)
    for v in VIDEOS7:
        for kind in ("static", "dynamic"):
            (root / "video" / f"{kind}__{v}.csv").write_bytes(g[f"{v}_{kind}_csv"].tobytes())
        (root / "audio_repeat_1" / MODEL7 / f"{v}.csv").write_bytes(g[f"{v}_audio_csv"].tobytes())
    fmt = tmp_path / "prediction_file_format.csv"
    fmt.write_text("image_location\n" + "\n".join(g["format_rows"].tolist()) + "\n")
    return str(root), str(fmt)


def test_weight_tables_are_the_reference_constants(golden):
    """get_weights_matrices.py:5-62: rows 0-6 of each matrix transposed = level-1 weights, row 7 = level-2 weights."""
    from avcer_amd import fusion as fu

    g = golden("dataset_fusion7")
    np.testing.assert_array_equal(np.array(fu.WEIGHTS_V_1), g["weights_1"][:7].T)
    np.testing.assert_array_equal(np.array(fu.WEIGHTS_V_2), g["weights_1"][7])
    np.testing.assert_array_equal(np.array(fu.WEIGHTS_AV7_1), g["weights_2"][:7].T)
    np.testing.assert_array_equal(np.array(fu.WEIGHTS_AV7_2), g["weights_2"][7])
    np.testing.assert_array_equal(np.array(fu.WEIGHTS_AV_1), g["weights_3"][:7].T)
    np.testing.assert_array_equal(np.array(fu.WEIGHTS_AV_2), g["weights_3"][7])


def test_seven_column_audio_tables_round_trip(golden, tmp_path):
    g = golden("dataset_fusion7")
    for v in VIDEOS7:
        p = iof.write_audio_csv(g[f"{v}_aud_rows"], g[f"{v}_aud_frames"], str(tmp_path / "w"), MODEL7, v)
        assert open(p, "rb").read() == g[f"{v}_audio_csv"].tobytes()
        rows, frames = iof.read_audio_csv(p)
        assert rows.shape[1] == 7 and np.array_equal(frames, g[f"{v}_aud_frames"])


@pytest.mark.gpu
def test_acl7_and_video_only_fusion_reproduce_the_reference_submissions(engine, golden, tmp_path):
    """get_pred_av.get_c_expr_db_pred with the Acl7 tables (7-column audio_repeat_1 CSVs, get_pred_av.py:362-365) and
    get_pred_video.get_c_expr_db_pred, single and double weights, both compound rules on and off: 32 files, byte for byte."""
    from avcer_amd import fusion as fu

    g = golden("dataset_fusion7")
    root, fmt = _materialise7(g, tmp_path)
    for weight_type, w2a, w2v in (("single", (1, 1, 1), (1, 1)), ("double", fu.WEIGHTS_AV7_2, fu.WEIGHTS_V_2)):
        for cwt in (False, True):
            for cm in (True, False):
                _, _, txt = iof.dataset_fusion(engine, fmt, root, ["video", "audio_repeat_1", MODEL7], list(VIDEOS7),
                                               fu.WEIGHTS_AV7_1, w2a, "AV_Acl7", weight_type, cwt, cm, save_path=str(tmp_path / "out"))
                key = f"{weight_type}_{int(cwt)}{int(cm)}"
                assert os.path.basename(txt) == f"C_EXPR_DB_AV_Acl7_sd_{weight_type}_{cwt}_{cm}.txt"
                assert open(txt, "rb").read() == g["av7_" + key].tobytes(), key
                _, _, paths = iof.dataset_fusion_video(engine, fmt, os.path.join(root, "video"), list(VIDEOS7), fu.WEIGHTS_V_1, w2v,
                                                       "V", weight_type, cwt, cm, save_path=str(tmp_path / "out"))
                assert os.path.basename(paths[1]) == f"C_EXPR_DB_V_static_{weight_type}_{cwt}_[True, False].txt"
                for kind, path in zip(("sd", "static", "dynamic"), paths):
                    assert open(path, "rb").read() == g[f"v_{kind}_{key}"].tobytes(), (kind, key)
