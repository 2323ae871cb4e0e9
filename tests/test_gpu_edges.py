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


def test_x3_activation_overflow_reaches_the_output_as_nan_and_is_counted(engine_static):
    """Range contract of AVCER_MODE_F16X3 (include/avcer_hip.h): activations are fp16 pairs, |x| < 65504.  Through the raw
    entry points a preprocessed tensor scaled far beyond that comes back as NaN probabilities -- never as finite, wrong
    ones -- AND the library's range-contract counter says why (avcer_x3_overflow_count > 0), while the f32 mode still
    computes it.  Inside the range the counter stays at 0."""
    eng = engine_static
    x = torch.from_numpy(synth.face_frames(7, 2)).float().permute(0, 3, 1, 2).contiguous() - 100.0
    eng.x3_overflow_count(reset=True)
    lg, pr, _ = eng.static_forward_nchw(x * 1.0e4, MODE_F16X3)
    assert torch.isnan(pr).all() and torch.isnan(lg).all()
    assert eng.x3_overflow_count(reset=True) > 0
    assert eng.x3_overflow_count(reset=True) == 0                     # the read above reset it
    lg32, pr32, _ = eng.static_forward_nchw(x * 1.0e4, MODE_FP32)
    assert torch.isfinite(lg32).all()
    assert eng.x3_overflow_count(reset=True) == 0                     # the f32 mode has no range limit and counts nothing
    # the same tensor inside the range: the two parity-grade modes agree, nothing is counted
    lg, pr, _ = eng.static_forward_nchw(x, MODE_F16X3)
    lg32, pr32, _ = eng.static_forward_nchw(x, MODE_FP32)
    assert torch.isfinite(pr).all() and (pr - pr32).abs().max() < 1e-4
    assert eng.x3_overflow_count(reset=True) == 0


def test_mirrors_repeat_an_overflowing_call_in_fp32(engine_static, engine_dynamic, sd_static, sd_dynamic):
    """What a caller of the drop-in mirrors sees when a checkpoint's activations leave fp16's range: numbers, the ones the
    reference's fp32 modules compute (get_prob_video.py:107-112) -- the call is repeated in MODE_FP32 (Engine.guarded)."""
    from avcer_amd.models import DynamicModel, StaticModel

    eng = engine_static
    x = (torch.from_numpy(synth.face_frames(7, 2)).float().permute(0, 3, 1, 2).contiguous() - 100.0) * 1.0e4
    ref_lg, ref_pr, ref_ft = eng.static_forward_nchw(x, MODE_FP32)
    eng.x3_overflow_count(reset=True)
    model = StaticModel(eng, sd_static)                                # default mode: MODE_F16X3
    before = eng.x3_fallbacks
    lg = model(x)
    assert eng.x3_fallbacks == before + 1
    assert torch.isfinite(lg).all() and torch.equal(lg, ref_lg) and torch.equal(model.activations["features"], ref_ft)
    # an in-range call is not repeated and stays in the fast mode
    x_ok = x * 1.0e-4
    lg_ok = model(x_ok)
    assert eng.x3_fallbacks == before + 1 and torch.equal(lg_ok, eng.static_forward_nchw(x_ok, MODE_F16X3)[0])
    # the LSTM mirror: windows far outside the range (the on-the-fly split of its f32 operand counts too)
    dyn = DynamicModel(eng, sd_dynamic)
    w = torch.full((2, 10, 512), 3.0e5)
    out = dyn(w)
    assert eng.x3_fallbacks == before + 2 and torch.equal(out, eng.dynamic_forward(w, MODE_FP32))


def test_nan_audio_window_is_not_an_overflow(engine_audio):
    """The reference's legitimate NaN -- the empty tail window under 'mean' padding (get_prob_audio_8_cl.py:78-92,
    data/utils.py:76-82) -- travels through every split site of the audio model and leaves the range-contract counter at
    0: a caller can tell it from an overflow."""
    eng = engine_audio
    wav = torch.from_numpy(synth.waveforms(5, 1, 16000)[0])
    eng.x3_overflow_count(reset=True)
    logits, lo, hi = audio_pipeline.audio_forward(eng, wav, 16000, 25, window=2, step=0.5, padding="mean", mode=MODE_F16X3)
    assert torch.isnan(logits[-1]).all() and torch.isfinite(logits[:-1]).all()
    assert eng.x3_overflow_count(reset=True) == 0
    # ... and a waveform scaled out of the range IS counted (the normaliser is off: the raw samples reach conv0 / LN / GELU)
    big = torch.from_numpy(synth.waveforms(6, 2, 32000)) * 1.0e7
    out = eng.audio_forward(big, normalize=False, mode=MODE_F16X3)
    n = eng.x3_overflow_count(reset=True)
    assert n > 0 or torch.isfinite(out).all()  # LayerNorm may pull the values back into range: then nothing broke


def test_guarded_call_is_not_charged_with_counts_left_by_earlier_work(engine_static, sd_static):
    """`Engine.guarded` owns the range-contract counter for its call: an overflow left behind by an earlier UNGUARDED raw call
    (or by a call that raised, or by abandoned side-stream work) must not force a needless fp32 repeat of an in-range call."""
    from avcer_amd.models import StaticModel

    eng = engine_static
    x = torch.from_numpy(synth.face_frames(7, 2)).float().permute(0, 3, 1, 2).contiguous() - 100.0
    eng.static_forward_nchw(x * 1.0e4, MODE_F16X3)                     # raw entry: leaves a count behind, nobody read it
    model = StaticModel(eng, sd_static)
    before = eng.x3_fallbacks
    lg = model(x)                                                      # in range
    assert eng.x3_fallbacks == before and torch.isfinite(lg).all()
    assert torch.equal(lg, eng.static_forward_nchw(x, MODE_F16X3)[0])
    assert eng.x3_overflow_count(reset=True) == 0


def test_bneck_chain_refuses_a_pass_whose_trunk_leaves_the_32_bit_offset_range(engine):
    """avcer_bneck_chain walks X / OUT [M][4 planes] with 32-bit byte offsets: 1400 frames of 55 x 55 at planes 64 is 4.3 GB of
    trunk (T1 alone, 1.08 GB, would pass).  AVCER_EINVAL before anything is launched -- the pointers are never dereferenced."""
    dev = engine.device
    dummy = torch.zeros(64, dtype=torch.int16, device=dev)
    b = torch.zeros(256, device=dev)
    for planes, nb, hw in ((64, 1400, 55), (128, 2750, 28)):
        with pytest.raises(AvcerError) as e:
            engine.bneck_chain(planes, nb, hw, hw, dummy, dummy, dummy, dummy, dummy, b, dummy, b, dummy, b)
        assert e.value.code == -1 and "4 GiB" in str(e.value)


def test_front_and_back_pass_sizes_do_not_show_in_any_result(engine_static):
    """avcer_set_static_batch / avcer_set_static_back_batch are scheduling knobs: 70 frames as front passes of 16 (the size the
    library defaults to for residency in the memory-side cache) feeding ONE back pass, as 7 + 70, and as the single-pass schedule."""
    eng = engine_static
    frames = torch.from_numpy(synth.face_frames(99, 70))
    try:
        eng.set_static_batch(1024, back=0)
        one = [t.cpu() for t in eng.static_forward(frames, MODE_F16X3)]
        for front, back in ((16, 2048), (7, 70), (32, 33), (1024, 2048)):
            eng.set_static_batch(front, back=back)
            got = [t.cpu() for t in eng.static_forward(frames, MODE_F16X3)]
            assert all(torch.equal(a, b) for a, b in zip(one, got)), (front, back)
        with pytest.raises(AvcerError):
            eng.set_static_batch(16, back=4096)
    finally:
        eng.set_static_batch(1024, back=0)


@pytest.mark.parametrize("n", [33, 256, 301, 1100])
def test_two_lane_split_of_mid_sized_calls_is_bit_identical(engine_static, n):
    """avcer_static_forward runs calls of 32-2048 frames as two half-batches on two HIP streams (api.hip static_forward_impl;
    BASELINE config 2 is 256 frames; 1100 frames: each lane runs its own front / back passes).  The split must not show: serial (avcer_set_static_lanes(1)) and two-lane results are
    equal bit for bit, an odd count splits 151 + 150, and the caller's stream sees the second lane's outputs (the join)."""
    eng = engine_static
    frames = torch.from_numpy(synth.face_frames(31, n)).to(eng.device)
    try:
        eng.set_static_lanes(1)
        one = [t.clone() for t in eng.static_forward(frames, MODE_F16X3)]
        eng.set_static_lanes(2)
        for _ in range(2):                               # the second call re-uses the lane stream and its events
            two = eng.static_forward(frames, MODE_F16X3)
            # consumed on the caller's stream right away, without a device synchronisation in between
            same = [torch.equal(a, b) for a, b in zip(one, two)]
            assert all(same), same
        m = min(n, 130)
        x32 = [t.clone() for t in eng.static_forward(frames[:m], MODE_FP32)]
        eng.set_static_lanes(1)
        assert all(torch.equal(a, b) for a, b in zip(x32, eng.static_forward(frames[:m], MODE_FP32)))
        # the range knob: a call outside it is serial (same bits, of course), bad ranges are refused
        eng.set_static_lanes(2, 2, 16)
        assert all(torch.equal(a, b) for a, b in zip(one, eng.static_forward(frames, MODE_F16X3)))
        with pytest.raises(AvcerError):
            eng.set_static_lanes(2, 64, 32)
        with pytest.raises(AvcerError):
            eng.set_static_lanes(3)
    finally:
        eng.set_static_lanes(2, 32, 2048)
