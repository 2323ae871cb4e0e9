"""Edge cases of the boundary: empty / degenerate inputs, error behaviour, single-element batches."""
import numpy as np
import pytest
import torch

from avcer_amd import audio_pipeline, fusion, synth, video_pipeline
from avcer_amd._lib import AvcerError
from avcer_amd.engine import MODE_F16X3, MODE_FP32

pytestmark = pytest.mark.gpu


def test_clip_without_any_face_yields_zero_tables(engine_static, engine_dynamic):
    clip = torch.from_numpy(synth.face_frames(1, 6))
    st, dy = video_pipeline.visual_forward(engine_static, clip, np.zeros(6, bool), 25)
    assert st.shape == (6, 7) and not st.any() and not dy.any()  # get_prob_video.py:175-178


def test_single_frame_and_single_window(engine_static, engine_dynamic, engine_audio):
    for mode in (MODE_FP32, MODE_F16X3):
        lg, pr, ft = engine_static.static_forward(torch.from_numpy(synth.face_frames(2, 1)), mode)
        assert lg.shape == (1, 7) and ft.shape == (1, 512) and abs(float(pr.sum()) - 1) < 1e-5
        out = engine_audio.audio_forward(torch.from_numpy(synth.waveforms(2, 1, 32000)), True, mode)
        assert out.shape == (1, 8) and torch.isfinite(out).all()
    st, dy = video_pipeline.visual_forward(engine_static, torch.from_numpy(synth.face_frames(2, 1)), [True], 25)
    assert st.shape == (1, 7) and dy.abs().sum() > 0  # frame 0 is an LSTM step: window = that feature x10


def test_argument_errors_are_reported_not_crashed(engine_static, engine_audio):
    with pytest.raises(ValueError):
        engine_static.static_forward(torch.zeros(2, 224, 224, 4, dtype=torch.uint8))
    with pytest.raises(ValueError):
        engine_static.static_forward_nchw(torch.zeros(1, 3, 200, 200))
    with pytest.raises(AvcerError) as e:
        engine_audio.audio_forward(torch.zeros(1, 300), True, MODE_FP32)  # shorter than the conv receptive field
    assert e.value.code == -1 and "too short" in str(e.value)
    with pytest.raises(AvcerError):
        engine_audio.audio_forward(torch.zeros(1, 100000), True, MODE_FP32)  # > 256 tokens
    with pytest.raises(AvcerError):
        engine_static.static_forward(torch.zeros(1, 224, 224, 3, dtype=torch.uint8), mode=7)


def test_models_must_be_loaded(sd_static):
    from avcer_amd.engine import Engine

    eng = Engine(0)
    with pytest.raises(AvcerError) as e:
        eng.static_forward(torch.zeros(1, 224, 224, 3, dtype=torch.uint8))
    assert e.value.code == -4  # AVCER_ESTATE
    with pytest.raises(AvcerError):
        eng.dynamic_forward(torch.zeros(1, 10, 512))
    eng.close()


def test_audio_padding_repeat_on_empty_tail_raises_like_reference(engine):
    wav = torch.from_numpy(synth.waveforms(5, 1, 16000)[0])
    with pytest.raises(ZeroDivisionError):  # data/utils.py:66 with an empty chunk
        audio_pipeline.audio_forward(engine, wav, 16000, 25, window=2, step=0.5, padding="repeat")


def test_fusion_without_audio_coverage_raises_like_reference(engine):
    with pytest.raises(IndexError):  # audio_df[-1] on an empty table, run.py:100
        fusion.fuse(engine, np.zeros((4, 7), np.float32), np.zeros((4, 7), np.float32), np.zeros((1, 8), np.float32),
                    [10], [12])


def test_c_abi_audio_chunks_empty_chunk_in_repeat_mode_is_nan_not_a_fault(engine):
    """The raw C entry (no host-side guard): an empty chunk in mode 2 (repeat) must not index with i % 0.  The row comes
    back as NaN, the other rows are untouched."""
    import ctypes as C

    wav = torch.arange(1, 101, dtype=torch.float32, device=engine.device)
    starts = torch.tensor([0, 50, 100], dtype=torch.int32, device=engine.device)
    ends = torch.tensor([10, 50, 100], dtype=torch.int32, device=engine.device)      # rows 1 and 2 are empty
    out = torch.zeros(3, 16, device=engine.device)
    rc = engine.lib.avcer_audio_chunks(engine.ctx, C.c_void_p(wav.data_ptr()), C.c_void_p(starts.data_ptr()),
                                       C.c_void_p(ends.data_ptr()), 3, 16, 2, C.c_void_p(out.data_ptr()), engine._stream())
    assert rc == 0
    torch.cuda.synchronize()
    o = out.cpu()
    assert o[0].tolist() == [float(1 + i % 10) for i in range(16)]
    assert torch.isnan(o[1]).all() and torch.isnan(o[2]).all()


def test_x3_activation_overflow_reaches_the_output_as_nan(engine_static):
    """Range contract of AVCER_MODE_F16X3 (include/avcer_hip.h): activations are fp16 pairs, |x| < 65504.  A preprocessed
    tensor scaled far beyond that must come back as NaN probabilities -- never as finite, wrong ones -- while the f32 mode
    still computes it."""
    x = torch.from_numpy(synth.face_frames(7, 2)).float().permute(0, 3, 1, 2).contiguous() - 100.0
    lg, pr, _ = engine_static.static_forward_nchw(x * 1.0e4, MODE_F16X3)
    assert torch.isnan(pr).all() and torch.isnan(lg).all()
    lg32, pr32, _ = engine_static.static_forward_nchw(x * 1.0e4, MODE_FP32)
    assert torch.isfinite(lg32).all()
    # the same tensor inside the range: the two parity-grade modes agree
    lg, pr, _ = engine_static.static_forward_nchw(x, MODE_F16X3)
    lg32, pr32, _ = engine_static.static_forward_nchw(x, MODE_FP32)
    assert torch.isfinite(pr).all() and (pr - pr32).abs().max() < 1e-4
