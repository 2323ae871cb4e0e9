#!/usr/bin/env python3
"""Does the grouped tile order (gm m-tiles x all n-tiles per group, gemm.hip launch_wd_t: gm = 8) explain the weights-direct
family's traffic above compulsory, and can it be re-tuned per layer?  (round-5 review item 6; profiles/r06_wd_traffic.txt says WHERE
the excess is: the audio encoder's linears at M = 12672 and stage 4.)  Lab builds of the library with other gm values, the
product's objects for every other unit; per variant the time of four layer shapes and, under rocprofv3 --pmc, their FETCH_SIZE.

    python tools/wd_gm_probe.py build                       # here: tools/lab/libavcer_gm<G>.so
    python tools/wd_gm_probe.py time                        # GPU: us per launch, every variant, alternating child processes
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/g<G> -- python3 tools/wd_gm_probe.py run <G>
    python tools/wd_gm_probe.py show out                    # read MB per launch by variant and shape
"""
import csv
import glob
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LAB = os.path.join(ROOT, "tools", "lab")
GMS = (1, 2, 4, 8, 16, 32, 128)
SHAPES = ("ffn2", "ffn1", "qkv", "l4c3")


def lib_of(g):
    return os.path.join(LAB, f"libavcer_gm{g}.so")


def build():
    from avcer_amd import build as b

    hipcc = b._hipcc()
    src = open(os.path.join(b.CSRC, "gemm.hip")).read()
    key = "    p.gm = 8;\n    p.ntn = p.N / 256;"
    assert src.count(key) == 1
    procs = []
    for g in GMS:
        d = os.path.join(LAB, f"gm{g}", "avcer_amd", "csrc")
        shutil.rmtree(os.path.join(LAB, f"gm{g}"), ignore_errors=True)
        os.makedirs(d)
        os.makedirs(os.path.join(LAB, f"gm{g}", "include"))
        shutil.copy(os.path.join(ROOT, "include", "avcer_hip.h"), os.path.join(LAB, f"gm{g}", "include"))
        for f in os.listdir(b.CSRC):
            if f.endswith((".hip", ".h")):
                shutil.copy(os.path.join(b.CSRC, f), d)
        open(os.path.join(d, "gemm.hip"), "w").write(src.replace(key, f"    p.gm = {g};\n    p.ntn = p.N / 256;"))
        procs.append((g, subprocess.Popen([hipcc] + b.FLAGS + ["-c", os.path.join(d, "gemm.hip"), "-o", os.path.join(d, "gemm.o")],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for g, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit(f"gm {g}: hipcc failed\n{out[-2000:]}")
        d = os.path.join(LAB, f"gm{g}", "avcer_amd", "csrc")
        objs = [os.path.join(d, "gemm.o")] + [os.path.join(b.CSRC, s.replace(".hip", ".o")) for s in b.SOURCES if s != "gemm.hip"]
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib_of(g)])
        shutil.rmtree(os.path.join(LAB, f"gm{g}"))
        print(lib_of(g))


def layers():
    from tools.layer_bench import conv2d, linear

    return {"ffn2": linear(12672, 4096, 1024, "ffn2 4096->1024 +res", res=True), "ffn1": linear(12672, 1024, 4096, "ffn1 1024->4096"),
            "qkv": linear(12672, 1024, 3072, "qkv 1024->3072"), "l4c3": conv2d(2048, 7, 512, 1, 1, 0, 2048, "l4.x.c3 512->2048 +res", res=True)}


def child(g, iters):
    from avcer_amd import _lib

    _lib.LIB = lib_of(g)
    from avcer_amd.engine import Engine
    from tools.layer_bench import run

    L = layers()
    run(Engine(0), [L[s] for s in SHAPES], 7, iters, f"gm {g}")


def show(d):
    print(f"{'gm':>4s} " + " ".join(f"{s + ' read MB':>16s}" for s in SHAPES))
    for g in GMS:
        fs = glob.glob(os.path.join(d, f"g{g}", "**", "*counter_collection.csv"), recursive=True)
        if not fs:
            continue
        rows = sorted((int(r["Start_Timestamp"]), float(r["Counter_Value"])) for r in csv.DictReader(open(fs[0]))
                      if r["Counter_Name"] == "FETCH_SIZE" and "conv_gemm_wd" in r["Kernel_Name"])
        per = len(rows) // len(SHAPES)  # launches per shape (2 warm-up + iters), in shape order
        vals = [sum(v for _, v in rows[i * per + 2:(i + 1) * per]) / max(per - 2, 1) * 2.0 * 1024 / 1e6 for i in range(len(SHAPES))]
        print(f"{g:4d} " + " ".join(f"{v:16.1f}" for v in vals))
    L = layers()
    print("compulsory read MB: " + ", ".join(f"{s} {(L[s]['in_elems'] + (L[s]['out_elems'] if L[s]['res'] else 0) + L[s]['d'].n * L[s]['d'].cin * L[s]['d'].kh * L[s]['d'].kw) * 4 / 1e6:.1f}" for s in SHAPES))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        child(int(sys.argv[2]), 3)
    elif sys.argv[1] == "child":
        child(int(sys.argv[2]), 20)
    elif sys.argv[1] == "show":
        show(sys.argv[2])
    else:
        for rnd in range(2):
            for g in GMS:
                subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(g)], check=True)
