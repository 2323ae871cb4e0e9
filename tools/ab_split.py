#!/usr/bin/env python3
"""Same-box A/B of the x3 mode's operand split: the product library (fp16 pairs) against the lab build with the round-3
bf16 pairs (tools/build_lab.sh), speed and error, one child process per library (two copies of the same symbols cannot
share a process), interleaved A B A B so that clock drift of the box hits both.

    bash tools/build_lab.sh && python tools/ab_split.py            # parent
    python tools/ab_split.py --lab tools/lab/some_one_off_build.so  # the same A/B against any other build of the library
    python tools/ab_split.py --child [--lib path.so]                # what the parent runs
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LAB = os.path.join(ROOT, "tools", "lab", "libavcer_hip_bf16split.so")


def child(lib):
    import numpy as np
    import torch

    from avcer_amd import _lib
    if lib:
        _lib.LIB = lib
    from avcer_amd import synth
    from avcer_amd.engine import MODE_F16X3, Engine
    from oracle import video as ov

    eng = Engine(0)
    sd = synth.static_state_dict(42)
    eng.load_static(sd)
    eng.load_audio(synth.audio_state_dict(42))
    frames8 = synth.face_frames(2468, 8)
    with torch.no_grad():
        ref = torch.softmax(ov.resnet50_forward(synth.to_torch(sd), ov.pth_processing(frames8))[0], 1).numpy()
    lg, pr, _ = eng.static_forward(torch.from_numpy(frames8), MODE_F16X3)
    out = {"lib": os.path.basename(lib or _lib.LIB), "static_max_dprob": float(np.abs(pr.cpu().numpy() - ref).max())}
    frames = torch.from_numpy(synth.face_frames(1, 2048)).to(eng.device)
    wav = torch.from_numpy(synth.waveforms(2, 128, 32000)).to(eng.device)

    def timeit(fn, iters):
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    out["static_2048_ms"] = timeit(lambda: eng.static_forward(frames, MODE_F16X3), 10)
    out["static_256_ms"] = timeit(lambda: eng.static_forward(frames[:256], MODE_F16X3), 20)
    out["audio_128_ms"] = timeit(lambda: eng.audio_forward(wav, True, MODE_F16X3), 10)
    out["mfma_ceiling_tflops"] = eng.measure_ceilings()[0]
    print("AB " + json.dumps(out), flush=True)


def main():
    if "--child" in sys.argv:
        lib = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else None
        return child(lib)
    lab = os.path.abspath(sys.argv[sys.argv.index("--lab") + 1]) if "--lab" in sys.argv else LAB  # any one-off build to compare with
    if not os.path.exists(lab):
        sys.exit(f"{lab} missing: run tools/build_lab.sh")
    rows = []
    for rep in range(2):
        for lib in (None, lab):
            cmd = [sys.executable, os.path.abspath(__file__), "--child"] + (["--lib", lib] if lib else [])
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=280)
            line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
            if r.returncode or not line:
                sys.exit(f"child failed ({r.returncode}): {r.stderr[-1500:]}")
            rows.append(json.loads(line[0][3:]))
            print(rows[-1], flush=True)
    for key in ("static_2048_ms", "static_256_ms", "audio_128_ms", "mfma_ceiling_tflops", "static_max_dprob"):
        a = [r[key] for r in rows if r["lib"] == "libavcer_hip.so"]
        b = [r[key] for r in rows if r["lib"] != "libavcer_hip.so"]
        print(f"{key:22s} product {min(a):10.4g}   {os.path.basename(lab)} {min(b):10.4g}   ratio {min(a) / min(b):.3f}")


if __name__ == "__main__":
    main()
