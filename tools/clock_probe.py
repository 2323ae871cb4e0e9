#!/usr/bin/env python3
"""Does the K loop's memory traffic cost CLOCK (power) rather than issue slots?  Runs one conv_gemm shape back to back for a
few seconds per library (product, lab ablations of tools/build_lab_wd.sh) and samples rocm-smi (sclk, socket power) meanwhile.

    python tools/clock_probe.py [--child lib|-]
"""
import glob
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
        d = json.loads(r.stdout)
        c = d[sorted(d)[0]]
        sclk = [v for k, v in c.items() if k.startswith("sclk clock speed")]
        pw = [v for k, v in c.items() if "Power" in k and "(W)" in k]
        return (int(re.sub(r"\D", "", sclk[0])) if sclk else -1, float(pw[0]) if pw else -1.0)
    except Exception as e:  # noqa: BLE001
        return (-1, -1.0)


def child(lib):
    import torch

    from avcer_amd import _lib
    if lib != "-":
        _lib.LIB = lib
    from avcer_amd.engine import Engine
    from tools.layer_bench import conv2d

    eng = Engine(0)
    L = conv2d(2048, 14, 256, 3, 1, 1, 256, "l3c2 2048f")
    d = L["d"]
    d.tile_n = 256
    m, k = d.batch * d.out_h * d.out_w, d.kh * d.kw * d.cin
    x = torch.randn(L["in_elems"] + 64, device=eng.device)
    w = eng.weight_frags(torch.randn(d.n, k, device=eng.device) / k ** 0.5)
    y = torch.empty(m * d.n + 64, device=eng.device)
    sc, bi = torch.ones(d.n, device=eng.device), torch.zeros(d.n, device=eng.device)
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.25)

    th = threading.Thread(target=poll)
    n = 0
    t0 = time.perf_counter()
    th.start()
    while time.perf_counter() - t0 < 5.0:
        for _ in range(50):
            eng.conv_gemm(d, 7, x, w, sc, bi, None, y)
        torch.cuda.synchronize()
        n += 50
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    good = [s for s in samples[2:] if s[0] > 0]
    print("CLK " + json.dumps({"us": dt / n * 1e6, "tf": 6.0 * m * d.n * k * n / dt / 1e12,
                               "sclk_mhz": sorted(s[0] for s in good)[len(good) // 2] if good else -1,
                               "power_w": sorted(s[1] for s in good)[len(good) // 2] if good else -1, "samples": len(good)}), flush=True)


def main():
    if "--child" in sys.argv:
        return child(sys.argv[sys.argv.index("--child") + 1])
    print("idle:", smi())
    libs = ["-"] + sorted(glob.glob(os.path.join(ROOT, "tools", "lab", "libavcer_hip_wd*.so")))
    for lib in libs:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], capture_output=True, text=True, timeout=120)
        line = [l for l in r.stdout.splitlines() if l.startswith("CLK ")]
        print(os.path.basename(lib), line[0] if line else ("FAILED " + r.stderr[-400:]), flush=True)


if __name__ == "__main__":
    main()
