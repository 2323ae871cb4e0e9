#!/usr/bin/env python3
"""In-kernel clock of the two hot MFMA kernels, the way MI355X_MICROARCH.md prescribes it ("DVFS give-back" item 6): a
separate DIAGNOSTIC build stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around the loop, the stamps leave
the kernel through a buffer nothing else reads, the kernel is launched back to back for >= 2 s on random data, and the clock
is  d(s_memtime) / d(s_memrealtime) x 100 MHz,  median over workgroups.  Board power and rocm-smi's sclk are NOT the test
(round 4's tools/clock_probe.py read those).

The diagnostic sources are PATCHED COPIES of csrc/gemm.hip and csrc/fused.hip written under tools/lab/clock_<arm>/ -- the
product sources carry no stamp and no ablation switch.  Arms of conv_gemm_wd_kernel (l3.x.c2, 2048 frames):
    product   the shipped K loop + stamps
    noW       without the weight-fragment loads (registers keep random fp16 bit patterns)
    noA       without the activation-tile DMA after all four ring slots have been filled once
    mfma      neither, and no LDS fragment reads: the MFMAs alone
and of bneck_kernel<64, 128, true, 0, false, 1> (1024 frames): stamps around the conv2 K loop and around the streaming loop.

    python3 tools/clock_lab.py build            # here (hipcc cross-compiles): tools/lab/libavcer_clock_<arm>.so
    python3 tools/clock_lab.py run [seconds]    # on the GPU box: one child process per arm, prints the table
"""
from __future__ import annotations

import os
import shutil
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LAB = os.path.join(ROOT, "tools", "lab")
ARMS = ("product", "noW", "noA", "mfma")
NSTAMP = 32768

STAMP_DECL = f"""
// ---- tools/clock_lab.py: diagnostic build only
__device__ unsigned long long g_lab_stamps[4 * {NSTAMP}];
extern "C" int avcer_lab_stamps(unsigned long long* host, int n) {{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lab_stamps), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}}
"""
STAMP = """    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long {c} = __builtin_amdgcn_s_memtime(), {r} = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
"""
RANDOM_FP16 = "((0x9E3779B9u * (unsigned)(lane * 4 + {k} + 17 * fn + 1)) & 0x8FFF8FFFu) | 0x30003000u"


def must(s: str, a: str, b: str, count: int = 1) -> str:
    assert s.count(a) == count, (s.count(a), a[:80])
    return s.replace(a, b)


def patch_gemm(src: str, arm: str) -> str:
    s = src
    s = must(s, "namespace {\n\nstruct GemmParams {", STAMP_DECL + "\nnamespace {\n\nstruct GemmParams {")
    # stamps around the K loop of the weights-direct kernel
    s = must(s, "    for (int s = 0; s < nk; s += 2) {  // nk is even",
             STAMP.format(c="lab_c0", r="lab_r0") + "    for (int s = 0; s < nk; s += 2) {  // nk is even")
    s = must(s, '    asm volatile("" : "+v"(ah[0]), "+v"(al[0]));  // the last step\'s look-ahead read',
             STAMP.format(c="lab_c1", r="lab_r1") +
             f"    if (tid == 0 && bid < {NSTAMP}) {{ g_lab_stamps[4 * bid] = lab_c1 - lab_c0; g_lab_stamps[4 * bid + 1] = lab_r1 - lab_r0; }}\n"
             '    asm volatile("" : "+v"(ah[0]), "+v"(al[0]));  // the last step\'s look-ahead read')
    if arm in ("noW", "mfma"):
        s = must(s, "    u32x4_t wh0[NFN], wl0[NFN], wh1[NFN], wl1[NFN];",
                 "    u32x4_t wh0[NFN], wl0[NFN], wh1[NFN], wl1[NFN];\n"
                 "    _Pragma(\"unroll\") for (int fn = 0; fn < NFN; ++fn) {  // lab arm: random finite fp16 pairs instead of loads\n"
                 "        wh0[fn] = u32x4_t{" + ", ".join(RANDOM_FP16.format(k=k) for k in range(4)) + "};\n"
                 "        wl0[fn] = wh0[fn] ^ 0x01230123u; wh1[fn] = wh0[fn] ^ 0x04560456u; wl1[fn] = wh0[fn] ^ 0x07890789u;\n"
                 "    }")
        a = s.index("#define AVCER_WD_LOAD_W1(WH, WL, FN)")
        b = s.index("#define AVCER_WD_LOAD_W(WH, WL)")
        s = s[:a] + '#define AVCER_WD_LOAD_W1(WH, WL, FN) do { asm volatile("" : "+v"(WH[FN]), "+v"(WL[FN])); } while (0)\n' + s[b:]
    if arm in ("noA", "mfma"):
        # every ring slot is filled once (T = 0 .. 3), then never again: the fragment reads keep returning random data
        s = must(s, "        const bool live_ = (T) < nk;                                                                                    \\\n",
                 "        const bool live_ = (T) < nk;                                                                                    \\\n"
                 "        if ((T) >= 4) break; /* lab arm: no activation DMA behind the first four tiles */                               \\\n")
    if arm == "mfma":
        a = s.index("#define AVCER_WD_READ(BASE, R, AH, AL)")
        b = s.index("#define AVCER_WD_MFMA(R, AH, AL, WH, WL)")
        s = s[:a] + '#define AVCER_WD_READ(BASE, R, AH, AL) do { asm volatile("" : "+v"(AH), "+v"(AL)); } while (0)\n' + s[b:]
        s = must(s, "    spx8_t ah[2], al[2];\n    AVCER_WD_ISSUE_A(0);",
                 "    spx8_t ah[2], al[2];\n"
                 "    {   // lab arm: random finite fp16 operands instead of LDS reads\n"
                 "        const int fn = 0;\n"
                 "        const u32x4_t r0 = u32x4_t{" + ", ".join(RANDOM_FP16.format(k=k + 5) for k in range(4)) + "};\n"
                 "        ah[0] = __builtin_bit_cast(spx8_t, r0); al[0] = __builtin_bit_cast(spx8_t, r0 ^ 0x00770077u);\n"
                 "        ah[1] = __builtin_bit_cast(spx8_t, r0 ^ 0x01010101u); al[1] = __builtin_bit_cast(spx8_t, r0 ^ 0x02220222u);\n"
                 "    }\n    AVCER_WD_ISSUE_A(0);")
    return s


def patch_fused(src: str) -> str:
    s = src
    s = must(s, "namespace {\n\n// a.w ~= ah.wh + ah.wl + al.wh", STAMP_DECL + "\nnamespace {\n\n// a.w ~= ah.wh + ah.wl + al.wh")
    # the plain (per-tap gather) conv2 loop: the form planes 64 runs
    s = must(s, "        for (int step = 0; step < NK; ++step) {\n            if (step + 1 < NK) issue(cur ^ 1);",
             STAMP.format(c="lab_c0", r="lab_r0").replace("    __b", "        __b").replace("    const", "        const") +
             "        for (int step = 0; step < NK; ++step) {\n            if (step + 1 < NK) issue(cur ^ 1);")
    s = must(s, "            pin(acc2);\n            __syncthreads();\n            cur ^= 1;\n        }\n    }\n    }\n",
             "            pin(acc2);\n            __syncthreads();\n            cur ^= 1;\n        }\n" +
             STAMP.format(c="lab_c1", r="lab_r1").replace("    __b", "        __b").replace("    const", "        const") +
             f"        if (threadIdx.x == 0 && blockIdx.x < {NSTAMP}) {{ g_lab_stamps[4 * blockIdx.x] = lab_c1 - lab_c0; g_lab_stamps[4 * blockIdx.x + 1] = lab_r1 - lab_r0; }}\n"
             "    }\n    }\n")
    s = must(s, "    for (int G = 0; G < NG; G += 2) {\n        group(G, rh[0], rl[0]);",
             STAMP.format(c="lab_c2", r="lab_r2") + "    for (int G = 0; G < NG; G += 2) {\n        group(G, rh[0], rl[0]);")
    s = must(s, "        group(G + 1, rh[1], rl[1]);\n    }\n",
             "        group(G + 1, rh[1], rl[1]);\n    }\n" + STAMP.format(c="lab_c3", r="lab_r3") +
             f"    if (threadIdx.x == 0 && blockIdx.x < {NSTAMP}) {{ g_lab_stamps[4 * blockIdx.x + 2] = lab_c3 - lab_c2; g_lab_stamps[4 * blockIdx.x + 3] = lab_r3 - lab_r2; }}\n")
    return s


def build():
    from avcer_amd import build as b

    hipcc = b._hipcc()
    gemm = open(os.path.join(b.CSRC, "gemm.hip")).read()
    fused = open(os.path.join(b.CSRC, "fused.hip")).read()
    procs = []
    for arm in ARMS + ("chain",):
        d = os.path.join(LAB, "clock_" + arm, "avcer_amd", "csrc")
        shutil.rmtree(os.path.join(LAB, "clock_" + arm), ignore_errors=True)
        os.makedirs(d)
        os.makedirs(os.path.join(LAB, "clock_" + arm, "include"))
        shutil.copy(os.path.join(ROOT, "include", "avcer_hip.h"), os.path.join(LAB, "clock_" + arm, "include"))
        for f in os.listdir(b.CSRC):
            if f.endswith((".hip", ".h")):
                shutil.copy(os.path.join(b.CSRC, f), d)
        if arm == "chain":
            open(os.path.join(d, "fused.hip"), "w").write(patch_fused(fused))
        else:
            open(os.path.join(d, "gemm.hip"), "w").write(patch_gemm(gemm, arm))
        objs = []
        for src in b.SOURCES:
            o = os.path.join(d, src.replace(".hip", ".o"))
            objs.append(o)
            procs.append((arm, subprocess.Popen([hipcc] + b.FLAGS + ["-c", os.path.join(d, src), "-o", o], stdout=subprocess.PIPE,
                                                stderr=subprocess.STDOUT, text=True)))
    for arm, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit(f"{arm}: hipcc failed\n{out[-3000:]}")
    for arm in ARMS + ("chain",):
        d = os.path.join(LAB, "clock_" + arm, "avcer_amd", "csrc")
        lib = os.path.join(LAB, f"libavcer_clock_{arm}.so")
        objs = [os.path.join(d, s.replace(".hip", ".o")) for s in b.SOURCES]
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
        shutil.rmtree(os.path.join(LAB, "clock_" + arm))
        print(lib)


def child(arm: str, seconds: float):
    import ctypes as C
    import time

    import torch
    from avcer_amd import _lib

    _lib.LIB = os.path.join(LAB, f"libavcer_clock_{arm}.so")
    from avcer_amd.engine import Engine
    from tools.layer_bench import conv2d

    eng = Engine(0)
    dev = eng.device
    eng.lib.avcer_lab_stamps.restype = C.c_int
    eng.lib.avcer_lab_stamps.argtypes = [C.c_void_p, C.c_int]
    torch.manual_seed(1)
    if arm == "chain":
        planes, nb, hw = 64, 1024, 55
        p4, M = 4 * planes, nb * hw * hw
        from avcer_amd import sp32
        t1 = sp32.to_sp32(torch.relu(torch.randn(M, planes, device=dev)))
        x = sp32.to_sp32(torch.relu(torch.randn(M, p4, device=dev)))
        out = torch.empty((M, 2 * p4), dtype=torch.int16, device=dev)
        t1n = torch.empty((M, 2 * planes), dtype=torch.int16, device=dev)
        w2 = eng.split_weight_rows(torch.randn(planes, 9 * planes, device=dev) * 0.05)
        w3 = eng.split_weight_rows(torch.randn(p4, planes, device=dev) * 0.1)
        w1 = eng.split_weight_rows(torch.randn(planes, p4, device=dev) * 0.05)
        b2, b3, b1 = torch.zeros(planes, device=dev), torch.zeros(p4, device=dev), torch.zeros(planes, device=dev)
        launch = lambda: eng.bneck_chain(planes, nb, hw, hw, t1, x, out, t1n, w2, b2, w3, b3, w1, b1)
        blocks = (M + 127) // 128
    else:
        L = conv2d(2048, 14, 256, 3, 1, 1, 256, "l3.x.c2 3x3 256")
        d = L["d"]
        m, k, n = d.batch * d.out_h * d.out_w, d.kh * d.kw * d.cin, d.n
        nel = L["in_elems"] + 64
        nel += (-nel) % 32
        xx = eng.split_weights(torch.relu(torch.randn(nel, device=dev)))
        w = eng.weight_frags(torch.randn(n, k, device=dev) / k ** 0.5)
        y = torch.empty((m * n + 64) * 2, device=dev, dtype=torch.int16)
        sc, bi = torch.ones(n, device=dev), torch.zeros(n, device=dev)
        launch = lambda: eng.conv_gemm(d, 7, xx, w, sc, bi, None, y)
        blocks = ((m + 127) // 128) * (n // 256)
    for _ in range(20):
        launch()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + seconds
    n_l = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while time.perf_counter() < t_end:  # >= `seconds` of back-to-back launches: the clock has settled under the load
        for _ in range(50):
            launch()
        n_l += 50
        torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    nst = min(blocks, NSTAMP)
    buf = (C.c_ulonglong * (4 * nst))()
    rc = eng.lib.avcer_lab_stamps(C.cast(buf, C.c_void_p), 4 * nst)
    assert rc == 0, rc
    a = list(buf)

    def clk(i):
        v = [a[4 * b + i] / a[4 * b + i + 1] * 100.0 for b in range(nst) if a[4 * b + i + 1] > 0]
        cyc = [a[4 * b + i] for b in range(nst) if a[4 * b + i + 1] > 0]
        return statistics.median(v), statistics.median(cyc), len(v)

    if arm == "chain":
        c1, cy1, n1 = clk(0)
        c2, cy2, n2 = clk(2)
        print(f"chain  bneck_kernel<64,128,true,0,false,1> 1024 frames: {us:8.1f} us/launch  conv2 loop {c1:7.1f} MHz ({cy1:.0f} cycles)  "
              f"streaming loop {c2:7.1f} MHz ({cy2:.0f} cycles)  [{n1} blocks, {n_l} launches in the settle phase]", flush=True)
    else:
        c, cy, n1 = clk(0)
        print(f"{arm:8s} conv_gemm_wd_kernel l3.x.c2 2048 frames: {us:8.1f} us/launch  K loop {c:7.1f} MHz ({cy:.0f} cycles per block)  "
              f"[{n1} blocks, {n_l} launches in the settle phase]", flush=True)


def run(seconds: float):
    print(f"# in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz around the loop, median over workgroups, after >= {seconds:g} s "
          "of back-to-back launches on random data; one process per arm (diagnostic builds: read the CLOCK and the cycle SHARES, "
          "not the launch times -- the stamps' fences forbid overlaps the product kernel has)", flush=True)
    for rep in range(2):
        for arm in ARMS + ("chain",):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", arm, str(seconds)], capture_output=True, text=True, timeout=300)
            out = [l for l in r.stdout.splitlines() if "MHz" in l]
            print(out[-1] if out else f"{arm}: FAILED rc={r.returncode}\n{r.stderr[-1500:]}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "child":
        child(sys.argv[2], float(sys.argv[3]))
    else:
        run(float(sys.argv[2]) if len(sys.argv) > 2 else 2.5)
