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
ARMS = ("product", "noW", "noA", "mfma", "epiNoStore", "epiNoMath")
NSTAMP = 32768

STAMP_DECL = f"""
// ---- tools/clock_lab.py: diagnostic build only
__device__ unsigned long long g_lab_stamps[4 * {NSTAMP}];
__device__ unsigned long long g_lab_stamps2[4 * {NSTAMP}];
__device__ unsigned long long g_lab_stamps3[4 * {NSTAMP}];  // entry / exit s_memrealtime, HW_ID | XCC_ID << 32, entry s_memtime
extern "C" int avcer_lab_stamps3(unsigned long long* host, int n) {{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lab_stamps3), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}}
extern "C" int avcer_lab_stamps2(unsigned long long* host, int n) {{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lab_stamps2), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}}
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
    # ... and at the first / last instruction of the kernel: prologue, loop and epilogue shares of a block's residency
    s = must(s, "    constexpr int BMT = 16 * NFM, BN = 256, NFN = 4, STAGES = 4, ABYTES = 128 * ROWB;",
             STAMP.format(c="lab_cs", r="lab_rs") + "    constexpr int BMT = 16 * NFM, BN = 256, NFN = 4, STAGES = 4, ABYTES = 128 * ROWB;")
    s = must(s, "    else wd_epilogue<OUT, 0, NFN, NFM>(p, acc, m_base, c0, lane, wmul);\n#endif",
             "    else wd_epilogue<OUT, 0, NFN, NFM>(p, acc, m_base, c0, lane, wmul);\n"
             '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have been accepted\n' +
             STAMP.format(c="lab_ce", r="lab_re") +
             f"    if (tid == 0 && bid < {NSTAMP}) {{ g_lab_stamps[4 * bid + 2] = lab_c0 - lab_cs; g_lab_stamps[4 * bid + 3] = lab_ce - lab_c1; }}\n#endif")
    if arm == "epiNoStore":  # the epilogue's arithmetic without its stores (the values are folded into one sink word per lane)
        s = must(s, "        char* yp = p.Y + sp32_byte(e);\n        *reinterpret_cast<uint4*>(yp) = make_uint4(h[0], h[1], h[2], h[3]);\n"
                    "        *reinterpret_cast<uint4*>(yp + 64) = make_uint4(l[0], l[1], l[2], l[3]);",
                 "        if ((h[0] ^ h[1] ^ h[2] ^ h[3] ^ l[0] ^ l[1] ^ l[2] ^ l[3]) == 0x12345678u) *reinterpret_cast<uint32_t*>(p.Y) = 1u;")
    if arm == "epiNoMath":  # the epilogue's stores without its arithmetic: raw accumulator bits go out
        s = must(s, "                finish8<OUT, ACT>(p, m, ch, scale_bias4(acc[2 * j][h + f], s0, b0), scale_bias4(acc[2 * j + 1][h + f], s1, b1), rr[f][0],\n"
                    "                                  rr[f][1], ovm);",
                 "                { char* yp_ = p.Y + sp32_byte(m * p.ldY + p.yoff + ch);\n"
                 "                  *reinterpret_cast<f32x4_t*>(yp_) = acc[2 * j][h + f]; *reinterpret_cast<f32x4_t*>(yp_ + 64) = acc[2 * j + 1][h + f]; }")
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
    # the spatial-tile form's conv2 loop (no DMA, weights by asm loads): only where the sources carry that form (the round-5
    # experiment archived in profiles/experiments/r05_bneck_t11_spatial_tile.diff.txt; arm "chain11")
    if "AVCER_T11_LOAD(0);" in s:
      s = must(s, "        AVCER_T11_LOAD(0);\n        AVCER_T11_LOAD(1);\n",
             STAMP.format(c="lab_c0", r="lab_r0").replace("    __b", "        __b").replace("    const", "        const") +
             "        AVCER_T11_LOAD(0);\n        AVCER_T11_LOAD(1);\n")
      s = must(s, "#undef AVCER_T11_LOAD\n#undef AVCER_T11_WAIT\n",
             "#undef AVCER_T11_LOAD\n#undef AVCER_T11_WAIT\n" +
             STAMP.format(c="lab_c1", r="lab_r1").replace("    __b", "        __b").replace("    const", "        const") +
             f"        if (threadIdx.x == 0 && blockIdx.x < {NSTAMP}) {{ g_lab_stamps[4 * blockIdx.x] = lab_c1 - lab_c0; g_lab_stamps[4 * blockIdx.x + 1] = lab_r1 - lab_r0; }}\n")
    # the resident-patch form's conv2 loop (planes 128; arm "chain128")
    s = must(s, "        for (int c = 0; c < NQ; ++c) {\n            // the previous chunk's last barrier freed the patch and both weight tiles",
             STAMP.format(c="lab_c0", r="lab_r0").replace("    __b", "        __b").replace("    const", "        const") +
             "        for (int c = 0; c < NQ; ++c) {\n            // the previous chunk's last barrier freed the patch and both weight tiles")
    s = must(s, "    } else {\n    unsigned a_off[A_ISS];",
             STAMP.format(c="lab_c1", r="lab_r1").replace("    __b", "        __b").replace("    const", "        const") +
             f"        if (threadIdx.x == 0 && blockIdx.x < {NSTAMP}) {{ g_lab_stamps[4 * blockIdx.x] = lab_c1 - lab_c0; g_lab_stamps[4 * blockIdx.x + 1] = lab_r1 - lab_r0; }}\n"
             "    } else {\n    unsigned a_off[A_ISS];")
    # kernel entry / exit, and one common place for the conv2-end stamp of either form (lab_cm: declared at function scope)
    s = must(s, "    float* sbias = reinterpret_cast<float*>(smem + TILES);\n    for (int i = threadIdx.x; i < NBIAS; i += 256)",
             STAMP.format(c="lab_cs", r="lab_rs") + "    unsigned long long lab_a0 = 0, lab_a1 = 0;\n"
             "    float* sbias = reinterpret_cast<float*>(smem + TILES);\n    for (int i = threadIdx.x; i < NBIAS; i += 256)")
    s = s.replace("g_lab_stamps[4 * blockIdx.x] = lab_c1 - lab_c0;", "lab_a0 = lab_c0; lab_a1 = lab_c1; g_lab_stamps[4 * blockIdx.x] = lab_c1 - lab_c0;")
    s = must(s, "    sp_commit(p.ovf, ovm);\n}\n\n\n// ------------------------------------------------------------------------------------------------ bottleneck tail",
             '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n' + STAMP.format(c="lab_ce", r="lab_re") +
             f"    if (threadIdx.x == 0 && blockIdx.x < {NSTAMP}) {{ g_lab_stamps2[4 * blockIdx.x] = lab_a0 - lab_cs; g_lab_stamps2[4 * blockIdx.x + 1] = lab_c2 - lab_a1; "
             "g_lab_stamps2[4 * blockIdx.x + 2] = lab_ce - lab_c3; g_lab_stamps2[4 * blockIdx.x + 3] = lab_ce - lab_cs; "
             "g_lab_stamps3[4 * blockIdx.x] = lab_rs; g_lab_stamps3[4 * blockIdx.x + 1] = lab_re; "
             "g_lab_stamps3[4 * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); }\n"
             "    sp_commit(p.ovf, ovm);\n}\n\n\n// ------------------------------------------------------------------------------------------------ bottleneck tail")
    s = must(s, "    for (int G = 0; G < NG; G += 2) {\n        group(G, rh[0], rl[0]);",
             STAMP.format(c="lab_c2", r="lab_r2") + "    for (int G = 0; G < NG; G += 2) {\n        group(G, rh[0], rl[0]);")
    s = must(s, "        group(G + 1, rh[1], rl[1]);\n    }\n",
             "        group(G + 1, rh[1], rl[1]);\n    }\n" + STAMP.format(c="lab_c3", r="lab_r3") +
             f"    if (threadIdx.x == 0 && blockIdx.x < {NSTAMP}) {{ g_lab_stamps[4 * blockIdx.x + 2] = lab_c3 - lab_c2; g_lab_stamps[4 * blockIdx.x + 3] = lab_r3 - lab_r2; }}\n")
    return s


def build(only=()):
    from avcer_amd import build as b

    hipcc = b._hipcc()
    gemm = open(os.path.join(b.CSRC, "gemm.hip")).read()
    fused = open(os.path.join(b.CSRC, "fused.hip")).read()
    procs = []
    for arm in ARMS + ("chain",):
        if only and arm not in only:
            continue
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
        if only and arm not in only:
            continue
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

    _lib.LIB = os.path.join(LAB, f"libavcer_clock_{'chain' if arm in ('chain11', 'chain128') else arm}.so")
    from avcer_amd.engine import Engine
    from tools.layer_bench import conv2d

    eng = Engine(0)
    dev = eng.device
    eng.lib.avcer_lab_stamps.restype = C.c_int
    eng.lib.avcer_lab_stamps.argtypes = [C.c_void_p, C.c_int]
    torch.manual_seed(1)
    t11 = arm == "chain11"
    p128 = arm == "chain128"
    if t11 or p128:
        arm = "chain"
        _lib.LIB = os.path.join(LAB, "libavcer_clock_chain.so")
    if arm == "chain":
        planes, nb, hw = (128, 1024, 28) if p128 else (64, 1024, 55)
        p4, M = 4 * planes, nb * hw * hw
        from avcer_amd import sp32
        t1 = sp32.to_sp32(torch.relu(torch.randn(M, planes, device=dev)))
        x = sp32.to_sp32(torch.relu(torch.randn(M, p4, device=dev)))
        out = torch.empty((M, 2 * p4), dtype=torch.int16, device=dev)
        t1n = torch.empty((M, 2 * planes), dtype=torch.int16, device=dev)
        w2f32 = torch.randn(planes, 9 * planes, device=dev) * 0.05
        w2 = eng.split_weight_rows(w2f32)
        kw = {"w2_frags": eng.weight_frags(w2f32)} if t11 else {}
        w3 = eng.split_weight_rows(torch.randn(p4, planes, device=dev) * 0.1)
        w1 = eng.split_weight_rows(torch.randn(planes, p4, device=dev) * 0.05)
        b2, b3, b1 = torch.zeros(planes, device=dev), torch.zeros(p4, device=dev), torch.zeros(planes, device=dev)
        launch = lambda: eng.bneck_chain(planes, nb, hw, hw, t1, x, out, t1n, w2, b2, w3, b3, w1, b1, **kw)
        blocks = nb * 25 if t11 else (M + 127) // 128
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
        eng.lib.avcer_lab_stamps2.restype = C.c_int
        eng.lib.avcer_lab_stamps2.argtypes = [C.c_void_p, C.c_int]
        buf2 = (C.c_ulonglong * (4 * nst))()
        assert eng.lib.avcer_lab_stamps2(C.cast(buf2, C.c_void_p), 4 * nst) == 0
        a2 = list(buf2)
        def dist(vals):
            v = sorted(vals)
            return f"p10 {v[len(v) // 10]:.0f} / p50 {v[len(v) // 2]:.0f} / p90 {v[len(v) * 9 // 10]:.0f} / mean {sum(v) / len(v):.0f}"
        segs = ("entry -> conv2 loop", "conv2 end -> streaming loop", "streaming end -> exit", "entry -> exit")
        print("    cycles per block (median): " + "; ".join(f"{n}: {dist([a2[4 * b + i] for b in range(nst)])}" for i, n in enumerate(segs))
              + "; conv2 loop: " + dist([a[4 * b] for b in range(nst)]) + "; streaming loop: " + dist([a[4 * b + 2] for b in range(nst)]), flush=True)
        eng.lib.avcer_lab_stamps3.restype = C.c_int
        eng.lib.avcer_lab_stamps3.argtypes = [C.c_void_p, C.c_int]
        buf3 = (C.c_ulonglong * (4 * nst))()
        assert eng.lib.avcer_lab_stamps3(C.cast(buf3, C.c_void_p), 4 * nst) == 0
        a3 = list(buf3)
        # blocks resident at once per CU: sweep the entry / exit real times of the blocks that ran on each (XCC, SE, CU)
        from collections import defaultdict
        per_cu = defaultdict(list)
        for b in range(nst):
            hw = a3[4 * b + 2]
            hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
            cu = (xcc, (hwid >> 13) & 0x7, (hwid >> 12) & 0x1, (hwid >> 8) & 0xf)  # XCC, SE_ID, SH_ID, CU_ID (gfx9 HW_ID layout)
            per_cu[cu].append((a3[4 * b], 1))
            per_cu[cu].append((a3[4 * b + 1], -1))
        tot_t, w_sum, hist, refill = 0, 0, defaultdict(int), []
        for ev in per_cu.values():
            ev.sort()
            cur, last, freed = 0, ev[0][0], []
            for t, d in ev:
                hist[cur] += t - last
                w_sum += cur * (t - last)
                tot_t += t - last
                cur, last = cur + d, t
                if d < 0:
                    freed.append(t)          # a slot of this CU became free ...
                elif freed:
                    refill.append((t - freed.pop(0)) * 0.01)  # ... and the next block entered it this many us later
        if os.environ.get("AVCER_LAB_TIMELINE"):  # the blocks of one CU in entry order: entry, exit (us from the CU's first entry)
            key = sorted(per_cu)[0]
            blocks_of = sorted((a3[4 * b], a3[4 * b + 1], b) for b in range(nst)
                               if ((a3[4 * b + 2] >> 32) & 0xf, ((a3[4 * b + 2] & 0xffffffff) >> 13) & 7, ((a3[4 * b + 2] & 0xffffffff) >> 12) & 1,
                                   ((a3[4 * b + 2] & 0xffffffff) >> 8) & 0xf) == key)
            t00 = blocks_of[0][0]
            print(f"    timeline of CU {key}: " + " ".join(f"[{(e - t00) * 0.01:.1f}-{(x - t00) * 0.01:.1f} b{bb}]" for e, x, bb in blocks_of[:45]), flush=True)
        refill.sort()
        if refill:
            print(f"    slot refill delay on a CU (block exit -> next block's entry, us): p10 {refill[len(refill) // 10]:.2f} / p50 {refill[len(refill) // 2]:.2f} / "
                  f"p90 {refill[len(refill) * 9 // 10]:.2f} / mean {sum(refill) / len(refill):.2f}", flush=True)
        print(f"    {len(per_cu)} distinct (XCC, SE, SH, CU) ids; resident blocks per CU, time-weighted: mean {w_sum / max(tot_t, 1):.2f}; share of time at 0 / 1 / 2 / 3 / 4+ blocks: "
              + " / ".join(f"{hist[k] / max(tot_t, 1):.2f}" for k in range(4)) + f" / {sum(v for k, v in hist.items() if k >= 4) / max(tot_t, 1):.2f}", flush=True)
        c1, cy1, n1 = clk(0)
        c2, cy2, n2 = clk(2)
        print(f"{'chain11 (spatial-tile form)' if t11 else 'chain128 (patch form, 28 x 28)' if p128 else 'chain  '} bneck_kernel<{planes},128,true,...> 1024 frames: {us:8.1f} us/launch  conv2 loop {c1:7.1f} MHz ({cy1:.0f} cycles)  "
              f"streaming loop {c2:7.1f} MHz ({cy2:.0f} cycles)  [{n1} blocks, {n_l} launches in the settle phase]", flush=True)
    else:
        c, cy, n1 = clk(0)
        pro = statistics.median(a[4 * b + 2] for b in range(nst))
        epi = statistics.median(a[4 * b + 3] for b in range(nst))
        slots = 512  # two blocks per CU
        resid = us * 1e-6 * c * 1e6 * slots / blocks  # cycles a block slot spends per block, dispatch gap included
        print(f"{arm:8s} conv_gemm_wd_kernel l3.x.c2 2048 frames: {us:8.1f} us/launch  K loop {c:7.1f} MHz  cycles per block: prologue {pro:.0f}, "
              f"K loop {cy:.0f}, epilogue {epi:.0f}; slot time per block {resid:.0f} (kernel time x 512 slots / {blocks} blocks)  "
              f"[{n1} blocks, {n_l} launches in the settle phase]", flush=True)


def run(seconds: float, arms=()):
    print(f"# in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz around the loop, median over workgroups, after >= {seconds:g} s "
          "of back-to-back launches on random data; one process per arm (diagnostic builds: read the CLOCK and the cycle SHARES, "
          "not the launch times -- the stamps' fences forbid overlaps the product kernel has)", flush=True)
    for rep in range(2):
        for arm in (arms or ARMS + ("chain",)):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", arm, str(seconds)], capture_output=True, text=True, timeout=300)
            out = [l for l in r.stdout.splitlines() if "MHz" in l or "cycles per block (median)" in l or "resident blocks" in l or "refill" in l or "timeline" in l]
            print("\n".join(out[-5:]) if out else f"{arm}: FAILED rc={r.returncode}\n{r.stderr[-1500:]}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(tuple(sys.argv[2:]))
    elif sys.argv[1] == "child":
        child(sys.argv[2], float(sys.argv[3]))
    else:
        run(float(sys.argv[2]) if len(sys.argv) > 2 else 2.5, tuple(sys.argv[3:]))
