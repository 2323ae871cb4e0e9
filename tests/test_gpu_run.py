"""`run.run_inference` end to end on the GPU (frames + detections + waveform -> compound predictions) against the same
chain assembled from the CPU oracle."""
import numpy as np
import pytest
import torch

from avcer_amd import run as arun
from avcer_amd import synth
from avcer_amd.engine import MODE_F16X3, MODE_FP32
from avcer_amd.fusion import WEIGHTS_AV_1
from oracle import audio as oa
from oracle import face as oface
from oracle import fusion as ofu
from oracle import video as ov
from test_face_cpu import G, golden_frames, golden_script

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine_all(engine, sd_static, sd_dynamic, sd_audio):
    engine.load_static(sd_static)
    engine.load_dynamic(sd_dynamic)
    engine.load_audio(sd_audio)
    return engine


@pytest.mark.parametrize("mode,tol", [(MODE_FP32, 1e-4), (MODE_F16X3, 1e-4)])
def test_run_inference_matches_oracle_chain(engine_all, sd_static, sd_dynamic, sd_audio, tmp_path, mode, tol):
    frames, script = golden_frames(), golden_script()
    total, fps, sr = len(frames), 25, 16000
    wav = synth.waveforms(99, 1, int(total / fps * sr))[0]
    out = arun.run_inference(engine_all, frames, wav, fps, detections=script, path_save_results=str(tmp_path),
                             name_video="clip_x", flag_save_prob=True, weights_prob_model=WEIGHTS_AV_1,
                             ce_weights_type=False, ce_mask=True, mode=mode)
    np.testing.assert_array_equal(out["records"], G["track_records"])
    # oracle chain
    recs, tiles = oface.process_video(frames, script)
    rows0 = np.where(recs[:, 1] == 0)[0]
    present = np.zeros(total, bool)
    present[recs[rows0, 0]] = True
    clip = np.zeros((total, 224, 224, 3), np.uint8)
    clip[recs[rows0, 0]] = tiles[rows0]
    st, dy = ov.visual_forward(sd_static, sd_dynamic, clip, present, fps, batched=True)
    a_rows, a_frames = oa.audio_forward(sd_audio, torch.from_numpy(wav), sr, fps, 4, 0.5, "mean")
    prob, am = ofu.fuse(st.astype(np.float32), dy.astype(np.float32), a_rows, a_frames, WEIGHTS_AV_1, (1, 1, 1), False, True)
    assert np.abs(out["static_probs"] - st).max() < tol
    assert np.abs(ofu.softmax(out["dynamic_logits"]) - ofu.softmax(dy.astype(np.float32))).max() < tol
    np.testing.assert_array_equal(out["audio_frames"], a_frames)
    assert np.abs(ofu.softmax(out["audio_rows"][:, :7]) - ofu.softmax(a_rows[:, :7])).max() < tol
    assert np.abs(out["compound_prob"] - prob).max() < tol
    for i, name in enumerate(("av", "vs", "vd", "a")):
        np.testing.assert_array_equal(out[name], am[i])
    assert (tmp_path / "static__clip_x.csv").exists() and (tmp_path / "audio" / "clip_x.csv").exists()
    assert 0 < out["real_time_factor"] < 60  # elapsed / video duration, as run.py:307 prints it


def test_run_inference_needs_a_first_track(engine_all):
    frames = golden_frames()[:2]
    with pytest.raises(FileNotFoundError):
        arun.run_inference(engine_all, frames, np.zeros(1600, np.float32), 25,
                           detections=[np.zeros((0, 15), np.float32)] * 2)


def test_failed_call_leaves_nothing_behind_for_the_next_one(engine_all):
    """The reference's failure path (no face track 00: os.listdir fails, get_prob_video.py:79) unwinds `run_inference` while its
    audio branch is already queued on the side stream.  The call joins that stream on the way out (avcer_amd/run.py), so the next
    video on the same engine starts clean: same results as on a fresh call, range-contract counter at 0, and a tracker error
    (zero-area detection: TypeError in the reference) surfaces as ValueError naming the frame, from the native tracker."""
    frames, script = golden_frames(), golden_script()
    total, fps, sr = len(frames), 25, 16000
    wav = synth.waveforms(99, 1, int(total / fps * sr))[0]
    ref = arun.run_inference(engine_all, frames, wav, fps, detections=script, mode=MODE_F16X3)
    for _ in range(2):
        with pytest.raises(FileNotFoundError):
            arun.run_inference(engine_all, frames, wav, fps, detections=[np.zeros((0, 15), np.float32)] * total, mode=MODE_F16X3)
    bad = [np.asarray(d).copy() for d in script]
    k = next(i for i, d in enumerate(bad) if len(d))
    bad[k][0, 2] = bad[k][0, 0]                                    # zero width: no track id in the reference
    with pytest.raises(ValueError, match=f"frame {k}"):
        arun.run_inference(engine_all, frames, wav, fps, detections=bad, mode=MODE_F16X3)
    assert engine_all.x3_overflow_count(reset=False) == 0
    again = arun.run_inference(engine_all, frames, wav, fps, detections=script, mode=MODE_F16X3)
    for key in ("static_probs", "dynamic_logits", "audio_rows", "compound_prob", "av", "records"):
        np.testing.assert_array_equal(again[key], ref[key])
