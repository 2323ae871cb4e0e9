"""The drop-in claim of INTEGRATION.md section 3, proven: the reference's own per-frame / per-window loops driven through
the mirror classes of avcer_amd/models.py (`StaticModel`, `DynamicModel`, `AudioModel`), one call per frame / window,
exactly the call sites the reference has --

    pth_model_static(x.to(device)) -> logits, activations["features"]    get_prob_video.py:103-115
    pth_model_dynamic(lstm_f.to(device))                                 get_prob_video.py:122-128
    self.processor(...)["input_values"][0]; self.audio_model(a_fss)      get_prob_audio_8_cl.py:87-93

-- against the tables the reference itself produced (tests/golden/visual_harness.npz, audio_model.npz, chunker rules)
and the CPU oracle.  The loops below restate the control flow of get_prob_video.py:91-178 and
get_prob_audio_8_cl.py:78-101 around those calls (no cv2 / pandas / ffmpeg: frames and the waveform are in memory).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from avcer_amd import synth
from avcer_amd.engine import MODE_F16X3, MODE_FP32
from avcer_amd.models import AudioModel, DynamicModel, StaticModel
from oracle import audio as oa
from oracle import video as ov

pytestmark = pytest.mark.gpu


def frame_loop(pth_model_static, pth_model_dynamic, activations, frames_u8, present, fps, device):
    """get_prob_video.py:77-187 with the crops in memory: what the reference does with its three module-level objects."""
    step = round((5 * fps) / 25)
    zeros = np.zeros((1, 7))
    last_output, lstm_features = None, []
    probs_static, probs_dynamic = [], []
    for idx in range(len(present)):
        if present[idx]:
            x = ov.pth_processing(frames_u8[idx:idx + 1])                      # data/utils.py:19-39 (restated: no torchvision here)
            with torch.no_grad():
                prediction = F.softmax(pth_model_static(x.to(device)), dim=1)
            output_s = prediction.clone().cpu().detach().numpy()
            if idx % step == 0:
                features = F.relu(activations["features"]).cpu().detach().numpy()
                lstm_features = [features] * 10 if len(lstm_features) == 0 else lstm_features[1:] + [features]
                lstm_f = torch.unsqueeze(torch.from_numpy(np.vstack(lstm_features)), 0)
                with torch.no_grad():
                    output_d = pth_model_dynamic(lstm_f.to(device)).cpu().detach().numpy()
                last_output = output_d
                activations.clear()
            else:
                output_d = last_output if last_output is not None else zeros
            probs_static.append(output_s[0])
            probs_dynamic.append(output_d[0])
        else:
            lstm_features = []
            if last_output is not None:
                probs_static.append(probs_static[-1])
                probs_dynamic.append(probs_dynamic[-1])
            else:
                probs_static.append(zeros[0])
                probs_dynamic.append(zeros[0])
    return np.array(probs_static), np.array(probs_dynamic)


@pytest.mark.parametrize("mode,tol_s,tol_d", [(MODE_FP32, 1e-4, 2e-5), (MODE_F16X3, 1e-4, 1e-4)])
@pytest.mark.parametrize("case", ["gap25", "gap30", "lead25", "full25"])
def test_reference_frame_loop_runs_unchanged_on_the_mirrors(engine, sd_static, sd_dynamic, golden, case, mode, tol_s, tol_d):
    g = golden("visual_harness")
    pth_model_static = StaticModel(engine, sd_static, mode=mode)       # INTEGRATION.md section 3, verbatim
    activations = pth_model_static.activations
    pth_model_dynamic = DynamicModel(engine, sd_dynamic, mode=mode)
    pth_model_static.to(engine.device).eval()
    pth_model_dynamic.to(engine.device).eval()
    clip = synth.face_frames(4321, 16)
    present, fps = g[f"{case}_present"], int(g[f"{case}_fps"])
    stat, dyn = frame_loop(pth_model_static, pth_model_dynamic, activations, clip, present, fps, engine.device)
    ref_s, ref_d = g[f"{case}_static"], g[f"{case}_dynamic"]
    assert stat.shape == ref_s.shape and dyn.shape == ref_d.shape
    assert stat.dtype == ref_s.dtype and dyn.dtype == ref_d.dtype      # float64 once a np.zeros placeholder row is stacked
    ds, dd = np.abs(stat - ref_s).max(), np.abs(dyn - ref_d).max()
    print(case, "mode", mode, "max|dprob static|", ds, "max|d dynamic logit|", dd)
    assert ds < tol_s and dd < tol_d
    assert (stat.argmax(1) == ref_s.argmax(1)).all() and (dyn.argmax(1) == ref_d.argmax(1)).all()


def test_static_mirror_batch_of_one_equals_row_of_a_batch(engine, sd_static):
    """One frame per call (the reference's pattern) and the same frame inside a batch: same bits, in both parity modes."""
    x = ov.pth_processing(synth.face_frames(4321, 5))
    for mode in (MODE_FP32, MODE_F16X3):
        m = StaticModel(engine, sd_static, mode=mode)
        whole = m(x.to(engine.device)).cpu()
        feats = m.activations["features"].cpu()
        for i in (0, 3):
            one = m(x[i:i + 1].to(engine.device)).cpu()
            assert torch.equal(one[0], whole[i]) and torch.equal(m.activations["features"].cpu()[0], feats[i])
        assert torch.equal(m.extract_features(x[:2].to(engine.device)).cpu(), feats[:2])


class _EmotionRecognitionLike:
    """The attributes `EmotionRecognition.load_audio_features` reads (get_prob_audio_8_cl.py:24-44), with the model swapped
    as INTEGRATION.md section 3 says."""

    def __init__(self, audio_model, processor, device, window, step, padding):
        self.audio_model, self.processor, self.device = audio_model, processor, device
        self.window, self.step, self.sr, self.padding = window, step, 16000, padding

    def load_audio_features(self, wav, fps):
        """get_prob_audio_8_cl.py:70-101 with the waveform in memory."""
        window_a = self.window * self.sr
        step_a = int(self.step * self.sr)
        probs, framess = [], []
        for start_a in range(0, len(wav) + 1, step_a):
            end_a = min(start_a + window_a, len(wav))
            chunk = wav[start_a:end_a]
            a_fss = oa.pad_wav(chunk, window_a) if self.padding == "repeat" else oa.pad_wav_zeros(chunk, window_a, mode=self.padding)
            a_fss = torch.unsqueeze(a_fss, 0)
            a_fss = self.processor(a_fss, sampling_rate=self.sr)["input_values"][0]
            a_fss = torch.from_numpy(np.asarray(a_fss))
            with torch.no_grad():
                prob = self.audio_model(a_fss.to(self.device))
            prob = prob.cpu().numpy()
            frames = [i for i in range(round(start_a / self.sr * fps), round(end_a / self.sr * fps + 1))]
            probs.extend([prob] * len(frames))
            framess.extend(frames)
        return np.array(probs), np.array(framess)


@pytest.mark.parametrize("mode,tol", [(MODE_FP32, 2e-5), (MODE_F16X3, 1e-4)])
def test_reference_audio_loop_runs_unchanged_on_the_mirror(engine, sd_audio, golden, mode, tol):
    from transformers import Wav2Vec2FeatureExtractor

    proc = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0, do_normalize=True,
                                    return_attention_mask=True)      # what AutoFeatureExtractor resolves to for this model
    er = _EmotionRecognitionLike(AudioModel(engine, sd_audio, mode=mode), proc, engine.device, window=4, step=1, padding="mean")
    wav = torch.from_numpy(synth.waveforms(5679, 1, 64000)[0])        # the clip of audio_model.npz's t64000 case
    fps = 25
    rows, frames = er.load_audio_features(wav, fps)
    ref_rows, ref_frames = oa.audio_forward(sd_audio, wav, 16000, fps, window=4, step=1, padding="mean", batched=False)
    assert rows.shape == ref_rows.shape == (len(ref_frames), 8) and rows.dtype == np.float32
    assert np.array_equal(frames, ref_frames)
    # the last window is the reference's empty tail chunk (64000 % 16000 == 0): NaN rows there, same frames
    nan_ref = np.isnan(ref_rows).all(axis=1)
    assert nan_ref.any() and np.array_equal(np.isnan(rows).all(axis=1), nan_ref)
    ok = ~nan_ref
    p, pr = oa_softmax(rows[ok, :7]), oa_softmax(ref_rows[ok, :7])
    print("audio loop mode", mode, "max|dlogit|", np.abs(rows[ok] - ref_rows[ok]).max(), "max|dprob|", np.abs(p - pr).max())
    assert np.abs(p - pr).max() < tol
    assert (rows[ok].argmax(1) == ref_rows[ok].argmax(1)).all()
    # first window = the whole 4 s clip: the logits the reference's ExprModelV3 itself produced (shape (8,) at batch 1)
    g = golden("audio_model")["t64000_logits"]
    first = rows[0]
    assert first.shape == g.shape == (8,)
    assert np.abs(oa_softmax(first[None, :7]) - oa_softmax(g[None, :7])).max() < tol


def oa_softmax(x):
    e = np.exp(x - x.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def test_static_call_larger_than_one_back_pass(engine, sd_static):
    """A single static call of more than 2048 frames (api.hip: the outer s0 loop over back passes): 2 back passes, 3 front
    passes.  Rows are compared with the same frames run in separate smaller calls -- bit-identical, since no kernel's
    accumulation order depends on the batch -- and a sample of rows with the CPU oracle."""
    engine.load_static(sd_static)
    n = 2049 + 16
    base = torch.from_numpy(synth.face_frames(77, 64)).to(engine.device)
    frames = base.repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()
    frames[2048:] = torch.from_numpy(synth.face_frames(78, n - 2048)).to(engine.device)   # the second back pass gets its own content
    for mode, tol in ((MODE_F16X3, 1e-4), (MODE_FP32, 1e-4)):
        lg, pr, ft = [t.cpu() for t in engine.static_forward(frames, mode)]
        assert torch.isfinite(lg).all()
        l2, p2, f2 = [t.cpu() for t in engine.static_forward(frames[2040:], mode)]      # rows 2040.. straddle the pass boundary
        assert torch.equal(lg[2040:], l2) and torch.equal(pr[2040:], p2) and torch.equal(ft[2040:], f2)
        l3, _, f3 = [t.cpu() for t in engine.static_forward(frames[:64], mode)]
        assert torch.equal(lg[:64], l3) and torch.equal(lg[64:128], l3) and torch.equal(ft[1984:2048], f3)
        idx = [0, 1023, 1024, 2047, 2048, n - 1]
        with torch.no_grad():
            ref, _ = ov.resnet50_forward(sd_static, ov.pth_processing(frames[idx].cpu().numpy()))
            ref = torch.softmax(ref, 1)
        d = (pr[idx] - ref).abs().max().item()
        print("2065-frame call, mode", mode, "max|dprob| vs oracle", d)
        assert d < tol


@pytest.mark.parametrize("mode", [MODE_FP32, MODE_F16X3])
def test_two_granularity_schedule_small_passes(engine, sd_static, mode):
    """avcer_set_static_batch(4) with 11 frames: front passes of 4 + 4 + 3 frames writing into the back buffer at offsets
    c0 > 0, back passes of 8 + 3 frames (s0 > 0).  Bit-identical to the single-pass result, and equal to the oracle."""
    engine.load_static(sd_static)
    frames = torch.from_numpy(synth.face_frames(4321, 11))
    try:
        engine.set_static_batch(1024)
        one = [t.cpu() for t in engine.static_forward(frames, mode)]
        engine.set_static_batch(4)
        many = [t.cpu() for t in engine.static_forward(frames, mode)]
    finally:
        engine.set_static_batch(1024)
    assert all(torch.equal(a, b) for a, b in zip(one, many))
    with torch.no_grad():
        lg, ft = ov.resnet50_forward(sd_static, ov.pth_processing(frames.numpy()))
    d = (many[1] - torch.softmax(lg, 1)).abs().max().item()
    print("two-granularity schedule, mode", mode, "max|dprob|", d, "max|dfeat|", (many[2] - ft).abs().max().item())
    assert d < 1e-4


def test_preprocess_video_and_predict_from_a_face_directory(engine, sd_static, sd_dynamic, tmp_path):
    """The file-level mirror of get_prob_video.preprocess_video_and_predict: JPEG face crops of one track on disk (a leading miss, a
    miss before the first LSTM evaluation and one after it) -> the two tables and the reference's CSV files; equal to the oracle's
    harness on the decoded arrays."""
    from PIL import Image

    from avcer_amd import io_formats, video_pipeline as vp

    engine.load_static(sd_static)
    engine.load_dynamic(sd_dynamic)
    rng = np.random.default_rng(3)
    d = tmp_path / "clip7" / "00"
    d.mkdir(parents=True)
    total = 9
    for i in range(total):
        if i in (0, 2, 6):
            continue
        Image.fromarray(rng.integers(0, 256, (120 + 7 * i, 100 + 3 * i, 3), dtype=np.uint8)).save(d / f"{i:06d}.jpg", quality=92)
    dyn, stat = vp.preprocess_video_and_predict(engine, str(tmp_path / "clip7"), str(tmp_path / "out"), fps=25, total_frames=total,
                                                flag_save_prob=True)
    frames, present = vp.read_face_dir(str(tmp_path / "clip7"), total)
    assert present.tolist() == [i not in (0, 2, 6) for i in range(total)]
    st_o, dy_o = ov.visual_forward(sd_static, sd_dynamic, frames, present, 25, batched=True)
    assert np.abs(stat - st_o).max() < 1e-4 and np.abs(dyn - dy_o).max() < 1e-3
    assert not stat[0].any() and not dyn[0].any()                      # before the first face: zero rows (get_prob_video.py:175-178)
    assert not stat[2].any()                                           # a miss before the first LSTM evaluation (frame 5): zeros too
    np.testing.assert_array_equal(stat[6], stat[5])                    # a miss after it holds the last rows (:168-172)
    np.testing.assert_array_equal(dyn[6], dyn[5])
    back_s = io_formats.read_visual_csv(str(tmp_path / "out" / "static__clip7.csv"))
    back_d = io_formats.read_visual_csv(str(tmp_path / "out" / "dynamic__clip7.csv"))
    np.testing.assert_allclose(back_s, stat, rtol=0, atol=1e-7)
    np.testing.assert_allclose(back_d, dyn, rtol=0, atol=1e-6)
