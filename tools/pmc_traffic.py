#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md 'HBM' section) into
HBM bytes per launch of the dominant kernel.

    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE [--out profiles/r02_traffic_x3.json] [--clips 128]
                                [--steps 2]   (steps = timed + warm-up steps of the profiled bench command)

Corrections applied exactly as the guide prescribes for gfx950: counter unit = KiB; FETCH_SIZE reports 1/2 of the bytes of
a 16-B-per-lane read stream (global_load and buffer_load...lds alike) -> doubled; WRITE_SIZE is exact for 16-B-per-lane
streaming stores.  Cross-check on a known byte count (the guide asks for one): bneck_kernel<64,128,true> reads T1 0.79 GB +
residual 3.17 GB = 3.96 GB per 1024-frame launch as compulsory bytes; raw FETCH_SIZE 2.33 GB, doubled 4.66 GB = 1.18 x the
compulsory bytes (the rest: 3x3 halo rows that miss the L2) -- so the factor 2 also holds for its 64-byte-segment loads.
"""
import csv
import glob
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MFMA_FAMILY = ("conv_gemm", "bneck_kernel", "bneck_tail2_kernel", "stem_pool")


def kernel_source_hash():
    """Same hash as bench.py and the library itself (avcer_amd/build.py): the figure is only valid for the kernels it was measured on."""
    sys.path.insert(0, ROOT)
    from avcer_amd.build import source_hash
    return source_hash()


def per_kernel(dirname, counter):
    f = glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        a = acc.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return acc


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    rows = []
    for name in fe:
        n, kib, secs = fe[name]
        wn, wkib, _ = wr.get(name, [0, 0.0, 0.0])
        rd = 2.0 * kib * 1024.0          # gfx950: FETCH_SIZE counts 64 B per 128-B request
        wb = wkib * 1024.0
        rows.append((rd + wb, name, n, rd, wb, secs, 2.0 * kib * 1024.0 + wb))
    rows.sort(reverse=True)
    print(f"{'kernel':70s} {'launches':>8s} {'read GB':>9s} {'write GB':>9s} {'B/launch':>12s} {'TB/s':>6s}")
    for tot, name, n, rd, wb, secs, _ in rows[:12]:
        print(f"{name[-70:]:70s} {n:8d} {rd/1e9:9.2f} {wb/1e9:9.2f} {tot/n:12.0f} {tot/secs/1e12 if secs else 0:6.2f}")
    gem = [r for r in rows if any(k in r[1] for k in MFMA_FAMILY)]
    tot = sum(r[0] for r in gem); n = sum(r[2] for r in gem); secs = sum(r[5] for r in gem)
    clips = int(sys.argv[sys.argv.index("--clips") + 1]) if "--clips" in sys.argv else 128
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 2
    everything = sum(r[0] for r in rows)
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        commit = None
    res = {"kernel": "MFMA kernels: conv_gemm*, bneck_kernel, bneck_tail2_kernel, stem_pool_kernel / stem_pool_u8_kernel (all instantiations)",
           "clips_per_gpu": clips, "commit": commit or None, "kernel_source_hash": kernel_source_hash(),
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 1 --warmup 1 --no-secondary --no-cpu --no-configs --no-events --no-overlap --one-lane",
           "profiled_steps": steps, "hbm_gb_per_step_mfma_kernels": tot / steps / 1e9, "hbm_gb_per_step_all_kernels": everything / steps / 1e9,
           "per_kernel": {r[1][-60:]: {"launches": r[2], "read_gb": r[3] / 1e9, "write_gb": r[4] / 1e9} for r in rows[:12]},
           "launches": n, "hbm_bytes_per_launch": tot / n,
           "read_bytes_per_launch": sum(r[3] for r in gem) / n, "write_bytes_per_launch": sum(r[4] for r in gem) / n,
           "hbm_tb_per_s_during_kernel": tot / secs / 1e12,
           "correction": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), KiB units, WRITE_SIZE as is"}
    print(json.dumps(res))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
