import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/avcer_amd') else os.getcwd())
import numpy as np, torch, time
from avcer_amd import _lib
if len(sys.argv) > 1: _lib.LIB = os.path.abspath(sys.argv[1])
from avcer_amd import synth
from avcer_amd.engine import Engine, MODE_F16X3, MODE_FP32
from oracle import audio as oa
eng = Engine(0)
wav = synth.waveforms(1357, 4, 32000)
for seed in (42, 43, 44):
    sd = synth.audio_state_dict(seed)
    eng.load_audio(sd)
    with torch.no_grad():
        ref = oa.expr_model_v3_forward(synth.to_torch(sd), torch.from_numpy(oa.normalize(wav))).numpy()
    for name, mode in (("fp32", MODE_FP32), ("x3", MODE_F16X3)):
        got = eng.audio_forward(torch.from_numpy(wav), True, mode).cpu().numpy()
        print(seed, name, "max|dlogit| %.3e" % np.abs(got - ref).max())
w = torch.from_numpy(synth.waveforms(2, 128, 32000)).cuda()
for _ in range(3): eng.audio_forward(w, True, MODE_F16X3)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): eng.audio_forward(w, True, MODE_F16X3)
torch.cuda.synchronize(); print("audio x3 128 windows: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
