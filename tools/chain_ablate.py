#!/usr/bin/env python3
"""What bounds the fused bottleneck chains (bneck_kernel, stages 1-2)?  Lab builds with pieces left out, timed in one process.

    python tools/chain_ablate.py build          # here (hipcc cross-compiles): tools/lab/libavcer_chain_<arm>.so
    python tools/chain_ablate.py run            # on the GPU box

Arms (patched copies of csrc/fused.hip; results of the patched kernels are wrong by construction, only their time is read):
    product   the shipped kernel
    noMFMA    every v_mfma of the file is a no-op (accumulators stay zero): all global / LDS-DMA traffic, the epilogue VALU and the
              stores remain -- the kernel's memory side alone
    noOUT     the OUT rows are computed but not stored            (1 KiB of the 2.5 KiB per position at planes 64)
    noRes     the residual rows of X are not loaded               (1 KiB)
    noT1N     the next block's conv1 output is not stored         (256 B)
    noHBM     noOUT + noRes + noT1N: only T1 is read              (256 B) -- the kernel's compute side alone
    trunk3    round 6: the CEILING of a 3-byte trunk (fp16 hi + 8-bit lo; the round-5 review's byte cut).  The residual rows are read and
              the OUT rows written at 3 bytes per element in a compact layout (row pitch 3 x 4P bytes: the hi halves of a row
              first, 16 bytes per lane, then its 8-bit lo halves, 8 bytes per lane; whole 64- / 32-byte pieces per lane group) with
              NO pack / unpack arithmetic: the byte traffic of the format at zero VALU cost.  A real pack costs ~ +4.5 VALU per
              stored value, an unpack ~ +2.5 per loaded one (exponent extraction, scale, convert, byte pack).

    python tools/chain_ablate.py build [arm ...]; python tools/chain_ablate.py run [arm ...]
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LAB = os.path.join(ROOT, "tools", "lab")
ARMS = ("product", "noMFMA", "noOUT", "noRes", "noT1N", "noHBM", "noW2", "noWS", "noPatch", "trunk3")
ONLY = tuple(a for a in sys.argv[2:] if a in ARMS) if len(sys.argv) > 2 and sys.argv[1] in ("build", "run") else ()
if ONLY:
    ARMS = ONLY


def must(s, a, b):
    assert s.count(a) == 1, (s.count(a), a[:70])
    return s.replace(a, b)


def patch(src, arm):
    s = src
    if arm == "noMFMA":
        s = must(s, '#include "gemm_dev.h"\n', '#include "gemm_dev.h"\n#define mfma_sp(a, b, c) (c)\n')
    if arm in ("noOUT", "noHBM"):
        s = must(s, "            if (m_ok[t]) {\n                char* yp = p.OUT + (size_t)(SUB == 1 ? x_row[t]",
                 "            if (m_ok[t] && p.M < 0) {\n                char* yp = p.OUT + (size_t)(SUB == 1 ? x_row[t]")
    if arm in ("noT1N", "noHBM"):
        s = must(s, "                if (m_ok[t]) {\n                    char* yp = p.T1N + (long)m_row[t] * (P * 4) + q * 128 + 16 * g;",
                 "                if (m_ok[t] && p.M < 0) {\n                    char* yp = p.T1N + (long)m_row[t] * (P * 4) + q * 128 + 16 * g;")
    if arm in ("noRes", "noHBM"):
        s = must(s, "                const char* rp = p.X + (size_t)x_row[t] + G * 128;\n                h[t] = *reinterpret_cast<const uint4*>(rp);\n"
                    "                l[t] = *reinterpret_cast<const uint4*>(rp + 64);\n",
                 "                const char* rp = p.X + (size_t)x_row[t] + G * 128;\n                h[t] = l[t] = make_uint4(0u, 0u, 0u, 0u);\n"
                 "                if (p.M < 0) { h[t] = *reinterpret_cast<const uint4*>(rp); l[t] = *reinterpret_cast<const uint4*>(rp + 64); }\n")
    if arm == "trunk3":
        s = must(s, "                char* yp = p.OUT + (size_t)(SUB == 1 ? x_row[t] : o_row[SUB > 1 ? t : 0]) + G * 128;\n"
                    "                *reinterpret_cast<spx8_t*>(yp) = oh[t];\n                *reinterpret_cast<spx8_t*>(yp + 64) = ol[t];\n",
                 "                char* yp = p.OUT + (size_t)m_row[t] * (4 * P * 3);\n"
                 "                *reinterpret_cast<spx8_t*>(yp + G * 64 + 16 * g) = oh[t];\n"
                 "                { const uint4 l4_ = __builtin_bit_cast(uint4, ol[t]); *reinterpret_cast<uint2*>(yp + 8 * P + G * 32 + 8 * g) = make_uint2(l4_.x, l4_.y); }\n")
        s = must(s, "                const char* rp = p.X + (size_t)x_row[t] + G * 128;\n                h[t] = *reinterpret_cast<const uint4*>(rp);\n"
                    "                l[t] = *reinterpret_cast<const uint4*>(rp + 64);\n",
                 "                const char* rp = p.X + (size_t)m_row[t] * (4 * P * 3);\n                h[t] = *reinterpret_cast<const uint4*>(rp + G * 64 + 16 * g);\n"
                 "                { const uint2 l2_ = *reinterpret_cast<const uint2*>(rp + 8 * P + G * 32 + 8 * g); l[t] = make_uint4(l2_.x, l2_.y, 0u, 0u); }\n")
    # arms of the spatial-tile form (run with the fragment copy): no conv2 weight loads after the first three K-steps, no
    # weight DMA in the streaming phase after group 0, no halo-patch DMA
    if arm == "noW2":
        s = must(s, "            if (ks + 3 < NKS) { AVCER_T11_LOAD(ks + 3); }", "            if (ks + 3 < NKS && p.M < 0) { AVCER_T11_LOAD(ks + 3); }")
        s = must(s, "            if (ks + 3 < NKS) AVCER_T11_WAIT(6, ks);\n            else if (ks + 2 < NKS) AVCER_T11_WAIT(4, ks);\n            else if (ks + 1 < NKS) AVCER_T11_WAIT(2, ks);\n            else AVCER_T11_WAIT(0, ks);",
                 "            AVCER_T11_WAIT(0, ks);")
    if arm == "noWS":
        s = must(s, "        if (G + 1 < NG) issue_group(G + 1);", "        if (G + 1 < NG && p.M < 0) issue_group(G + 1);")
    if arm == "noPatch":
        s = must(s, "                if (ii < PIECES) dma16(t1rs, smem + q * (T11_SLOTS * ROWB) + ii * 1024, off);",
                 "                if (ii < PIECES && p.M < 0) dma16(t1rs, smem + q * (T11_SLOTS * ROWB) + ii * 1024, off);")
    return s


def build():
    from avcer_amd import build as b

    hipcc = b._hipcc()
    fused = open(os.path.join(b.CSRC, "fused.hip")).read()
    procs = []
    for arm in ARMS:
        d = os.path.join(LAB, "chain_" + arm, "avcer_amd", "csrc")
        shutil.rmtree(os.path.join(LAB, "chain_" + arm), ignore_errors=True)
        os.makedirs(d)
        os.makedirs(os.path.join(LAB, "chain_" + arm, "include"))
        shutil.copy(os.path.join(ROOT, "include", "avcer_hip.h"), os.path.join(LAB, "chain_" + arm, "include"))
        for f in os.listdir(b.CSRC):
            if f.endswith((".hip", ".h")):
                shutil.copy(os.path.join(b.CSRC, f), d)
        open(os.path.join(d, "fused.hip"), "w").write(patch(fused, arm))
        for src in b.SOURCES:
            o = os.path.join(d, src.replace(".hip", ".o"))
            if src != "fused.hip" and os.path.exists(os.path.join(b.CSRC, src.replace(".hip", ".o"))):
                shutil.copy(os.path.join(b.CSRC, src.replace(".hip", ".o")), o)   # unpatched units: the product's objects
                continue
            procs.append((arm, subprocess.Popen([hipcc] + b.FLAGS + ["-c", os.path.join(d, src), "-o", o], stdout=subprocess.PIPE,
                                                stderr=subprocess.STDOUT, text=True)))
    for arm, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit(f"{arm}: hipcc failed\n{out[-3000:]}")
    for arm in ARMS:
        d = os.path.join(LAB, "chain_" + arm, "avcer_amd", "csrc")
        lib = os.path.join(LAB, f"libavcer_chain_{arm}.so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(d, s.replace(".hip", ".o")) for s in b.SOURCES] + ["-o", lib])
        shutil.rmtree(os.path.join(LAB, "chain_" + arm))
        print(lib)


def child(arm):
    import torch
    from avcer_amd import _lib

    _lib.LIB = os.path.join(LAB, f"libavcer_chain_{arm}.so")
    from avcer_amd.engine import Engine

    eng = Engine(0)
    dev = eng.device
    res = []
    for planes, hw in ((64, 55), (128, 28)):
        nb, p4 = 1024, 4 * planes
        M = nb * hw * hw
        g = torch.Generator(device=dev).manual_seed(1)
        t1 = eng.split_weights(torch.relu(torch.randn(M * planes, device=dev, generator=g))).view(M, -1)
        x = eng.split_weights(torch.relu(torch.randn(M * p4, device=dev, generator=g))).view(M, -1)
        out, t1n = torch.empty_like(x), torch.empty_like(t1)
        w2f32 = torch.randn(planes, 9 * planes, device=dev) * 0.05
        w2 = eng.split_weight_rows(w2f32)
        w3 = eng.split_weight_rows(torch.randn(p4, planes, device=dev) * 0.1)
        w1 = eng.split_weight_rows(torch.randn(planes, p4, device=dev) * 0.05)
        b2, b3, b1 = torch.zeros(planes, device=dev), torch.zeros(p4, device=dev), torch.zeros(planes, device=dev)
        kw = {"w2_frags": eng.weight_frags(w2f32)} if planes == 64 else {}   # planes 64: the shipped spatial-tile form
        call = lambda: eng.bneck_chain(planes, nb, hw, hw, t1, x, out, t1n, w2, b2, w3, b3, w1, b1, **kw)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        res.append(sorted(ts)[2])
        del t1, x, out, t1n
    print(f"{arm:8s}  planes 64 (1024 x 55 x 55, middle block, spatial-tile form): {res[0]:8.1f} us    planes 128 (1024 x 28 x 28): {res[1]:8.1f} us", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "child":
        child(sys.argv[2])
    else:  # one child process per arm (a library is loaded once per process), back to back on the same GPU, two rounds
        for rnd in range(2):
            for arm in ARMS:
                subprocess.run([sys.executable, os.path.abspath(__file__), "child", arm], check=True)
