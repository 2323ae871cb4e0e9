#!/usr/bin/env python3
"""A few passes of the two big forward calls (static CNN on 2048 frames, audio model on 128 windows of 2 s) for a
`rocprofv3 --kernel-trace --stats` A/B of two library builds:

    rocprofv3 --kernel-trace --stats --output-format csv -d out -o x -- python3 tools/kstats_run.py [path/to/libavcer_hip.so]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from avcer_amd import _lib
    _lib.LIB = os.path.abspath(sys.argv[1])
import torch  # noqa: E402
from avcer_amd import synth  # noqa: E402
from avcer_amd.engine import MODE_F16X3, Engine  # noqa: E402

eng = Engine(0)
eng.load_static(synth.static_state_dict(42))
eng.load_audio(synth.audio_state_dict(42))
frames = torch.from_numpy(synth.face_frames(1234, 2048)).cuda()
wav = torch.from_numpy(synth.waveforms(5678, 128, 32000)).cuda()
for _ in range(6):
    eng.static_forward(frames, MODE_F16X3)
    eng.audio_forward(wav, True, MODE_F16X3)
    torch.cuda.synchronize()
