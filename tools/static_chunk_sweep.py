"""Time the static CNN (1280 frames, as in the bench step) for depth-first chunk sizes; one process per setting."""
import os
import subprocess
import sys

CHILD = r'''
import sys, time, torch
sys.path.insert(0, ".")
from avcer_amd import synth
from avcer_amd.engine import Engine
from avcer_amd.models import StaticModel
mode = int(sys.argv[1])
eng = Engine(0)
m = StaticModel(eng, synth.static_state_dict(1), mode=mode)
x = torch.from_numpy(synth.face_frames(3, 64)).cuda().repeat(20, 1, 1, 1)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    p = eng.static_forward(x, mode)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    p = eng.static_forward(x, mode)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("RESULT %.3f ms  checksum %.6f" % (dt * 1e3, float(p[1].double().sum())))
'''

def main():
    modes = [int(a) for a in sys.argv[1:]] or [2]
    for mode in modes:
        for front in (1, 2, 3):
            for chunk in (0, 16, 32, 64, 128):
                if chunk == 0 and front != 1:
                    continue
                env = dict(os.environ, AVCER_STATIC_CHUNK=str(chunk), AVCER_STATIC_FRONT=str(front))
                r = subprocess.run([sys.executable, "-c", CHILD, str(mode)], env=env, capture_output=True, text=True, timeout=300)
                line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
                print(f"mode {mode} front {front} chunk {chunk:4d}: {line[0] if line else 'FAILED ' + r.stderr[-400:]}", flush=True)

if __name__ == "__main__":
    main()
