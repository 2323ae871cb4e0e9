#!/usr/bin/env python3
"""What the kernels of the hot path cost in CLOCK: each case runs back to back for a few seconds while rocm-smi is sampled
(sclk, socket power).  Finding of round 4 (profiles/r04_clock_probe.txt): every MFMA-heavy kernel runs AT the socket power
cap; the chip answers extra memory traffic with a lower clock, not with stalls one could schedule away.

    python tools/clock_probe.py [seconds per case] [--lib one-off-build.so] [--only substring]
"""
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10)
        c = json.loads(r.stdout)
        c = c[sorted(c)[0]]
        sclk = [v for k, v in c.items() if k.startswith("sclk clock speed")]
        pw = [v for k, v in c.items() if "Power" in k and "(W)" in k]
        return (int(re.sub(r"\D", "", sclk[0])) if sclk else -1, float(pw[0]) if pw else -1.0)
    except Exception:  # noqa: BLE001
        return (-1, -1.0)


def main():
    import torch

    if "--lib" in sys.argv:  # a one-off experiment build of the library
        from avcer_amd import _lib
        _lib.LIB = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    from avcer_amd import synth
    from avcer_amd.engine import MODE_BF16, MODE_F16X3, MODE_FP32, Engine
    from avcer_amd.sp32 import to_sp32
    from tools.layer_bench import conv2d, linear

    argv = list(sys.argv[1:])
    only = argv.pop(argv.index("--only") + 1) if "--only" in argv else ""
    secs = float(argv[0]) if argv and not argv[0].startswith("--") else 4.0
    eng = Engine(0)
    dev = eng.device
    eng.load_static(synth.static_state_dict(42))
    eng.load_audio(synth.audio_state_dict(42))
    frames = torch.from_numpy(synth.face_frames(1, 2048)).to(dev)
    wav = torch.from_numpy(synth.waveforms(2, 128, 32000)).to(dev)
    cases = []

    def gemm_case(L, label):
        d = L["d"]
        d.tile_n = 256
        m, k = d.batch * d.out_h * d.out_w, d.kh * d.kw * d.cin
        x = to_sp32(torch.relu(torch.randn(L["in_elems"] // d.cin, d.cin, device=dev))).reshape(-1)
        w = eng.weight_frags(torch.randn(d.n, k, device=dev) / k ** 0.5)
        y = torch.empty(m * d.n * 2 + 64, dtype=torch.int16, device=dev)
        sc, bi = torch.ones(d.n, device=dev), torch.zeros(d.n, device=dev)
        cases.append((label, lambda: eng.conv_gemm(d, 7, x, w, sc, bi, None, y), 6.0 * m * d.n * k))

    def chain_case(planes, nb, hw, label):
        p4, M = 4 * planes, nb * hw * hw
        t1 = to_sp32(torch.relu(torch.randn(M, planes, device=dev)))
        x = to_sp32(torch.relu(torch.randn(M, p4, device=dev)))
        out, t1n = torch.empty_like(x), torch.empty_like(t1)
        w2 = eng.split_weight_rows(torch.randn(planes, 9 * planes, device=dev) * 0.05)
        w3 = eng.split_weight_rows(torch.randn(p4, planes, device=dev) * 0.1)
        w1 = eng.split_weight_rows(torch.randn(planes, p4, device=dev) * 0.05)
        b2, b3, b1 = torch.zeros(planes, device=dev), torch.zeros(p4, device=dev), torch.zeros(planes, device=dev)
        cases.append((label, lambda: eng.bneck_chain(planes, nb, hw, hw, t1, x, out, t1n, w2, b2, w3, b3, w1, b1),
                      6.0 * M * planes * planes * 17))

    gemm_case(conv2d(2048, 14, 256, 3, 1, 1, 256, ""), "wd l3c2 3x3 256, 2048 frames")
    gemm_case(linear(12672, 4096, 1024, ""), "wd ffn2 4096->1024, 128 windows")
    chain_case(64, 1024, 55, "chain planes 64, 1024 frames")
    chain_case(128, 2048, 28, "chain planes 128, 2048 frames")
    cases.append(("static CNN x3, 2048 frames", lambda: eng.static_forward(frames, MODE_F16X3), 6.927e9 * 2048 * 3))
    cases.append(("static CNN fp32, 512 frames", lambda: eng.static_forward(frames[:512], MODE_FP32), 6.927e9 * 512))
    cases.append(("static CNN bf16, 2048 frames", lambda: eng.static_forward(frames, MODE_BF16), 6.927e9 * 2048))
    cases.append(("audio model x3, 128 windows", lambda: eng.audio_forward(wav, True, MODE_F16X3), 44.26e9 * 128 * 3))
    cases.append(("register-only MFMA loop + 1 GiB copies (measure_ceilings)", lambda: eng.measure_ceilings(), 0.0))
    print("idle: sclk %d MHz, %.0f W" % smi())
    for label, fn, flop in cases:
        if only not in label:
            continue
        samples, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                samples.append(smi())
                time.sleep(0.2)

        fn(); fn()
        torch.cuda.synchronize()
        th = threading.Thread(target=poll)
        n, t0 = 0, time.perf_counter()
        th.start()
        while time.perf_counter() - t0 < secs:
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
            n += 4
        dt = time.perf_counter() - t0
        stop.set()
        th.join()
        good = sorted(s for s in samples[2:] if s[0] > 0)
        clk = good[len(good) // 2][0] if good else -1
        pw = sorted(s[1] for s in good)[len(good) // 2] if good else -1
        print(f"{label:62s} {dt / n * 1e3:9.3f} ms  {flop * n / dt / 1e12:7.0f} TF executed   sclk {clk:5d} MHz  {pw:6.0f} W  ({len(good)} samples)", flush=True)


if __name__ == "__main__":
    main()
