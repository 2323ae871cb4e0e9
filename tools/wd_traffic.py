#!/usr/bin/env python3
"""Where the MFMA kernels' HBM traffic exceeds the compulsory bytes, launch by launch (round-5 review item 6: "why is the
weights-direct family at 1.48 x compulsory?").

    python3 tools/wd_traffic.py log gpurun_out/x/launches.json                      the launch list of one step (library's own log)
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/x/f -- python3 tools/wd_traffic.py run
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/x/w -- python3 tools/wd_traffic.py run
    python3 tools/wd_traffic.py show gpurun_out/x/launches.json gpurun_out/x/f gpurun_out/x/w

The step's launch sequence is deterministic, so the last len(list) MFMA dispatches of a counter pass are the launches of the
list, in order.  Counter corrections as in tools/pmc_traffic.py (FETCH_SIZE x 2 on gfx950, KiB units).  FETCH_SIZE counts
memory-side (Infinity Cache) hits as traffic: a re-read of a tensor a previous launch just wrote is in the figure."""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MFMA = ("conv_gemm", "bneck_kernel", "bneck_tail2_kernel", "stem_pool")


def pipeline():
    import torch
    from avcer_amd import synth
    from avcer_amd.engine import MODE_F16X3
    from avcer_amd.pipeline import AVPipeline

    pipe = AVPipeline(device=0, seed=42, mode=MODE_F16X3)
    pipe.overlap_branches = False
    frames = torch.from_numpy(synth.face_frames(1234, 128 * 16)).reshape(128, 16, 224, 224, 3).cuda()
    wav = torch.from_numpy(synth.waveforms(5678, 128, 32000)).cuda()
    return pipe, frames, wav, torch


def run(log=None):
    pipe, frames, wav, torch = pipeline()
    for _ in range(2):
        pipe.run_clips(frames, wav, 25)
    torch.cuda.synchronize()
    if log:
        pipe.engine.profile_enable(True)
    pipe.run_clips(frames, wav, 25)
    torch.cuda.synchronize()
    if log:
        json.dump(pipe.engine.profile_read_launches(), open(log, "w"))
        pipe.engine.profile_enable(False)


def dispatches(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in MFMA):
                rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def show(log, fdir, wdir):
    launches = json.load(open(log))
    fe, wr = dispatches(fdir, "FETCH_SIZE"), dispatches(wdir, "WRITE_SIZE")
    n = len(launches)
    fe, wr = fe[-n:], wr[-n:]
    assert len(fe) == n and len(wr) == n, (n, len(fe), len(wr))
    rows = []
    for l, (_, kn, fkib), (_, kn2, wkib) in zip(launches, fe, wr):
        key = l["family"].replace("stem_pool_kernel", "stem_pool")
        assert key in kn and key in kn2, (l, kn, kn2)
        rd, wb = 2.0 * fkib * 1024.0, wkib * 1024.0
        rows.append(dict(l, read=rd, written=wb, ratio=(rd + wb) / l["bytes"]))
    print(f"{'#':>3s} {'family':24s} {'M':>8s} {'N':>5s} {'K':>5s} {'ms':>7s} {'compulsory MB':>14s} {'read MB':>9s} {'written MB':>11s} {'ratio':>6s}")
    for i, r in enumerate(rows):
        print(f"{i:3d} {r['family']:24s} {r['m']:8d} {r['n']:5d} {r['k']:5d} {r['ms']:7.3f} {r['bytes'] / 1e6:14.1f} {r['read'] / 1e6:9.1f} "
              f"{r['written'] / 1e6:11.1f} {r['ratio']:6.2f}")
    print("\nby family:")
    for fam in sorted({r["family"] for r in rows}):
        rs = [r for r in rows if r["family"] == fam]
        c, t = sum(r["bytes"] for r in rs), sum(r["read"] + r["written"] for r in rs)
        print(f"  {fam:24s} {len(rs):3d} launches  compulsory {c / 1e9:7.2f} GB  measured {t / 1e9:7.2f} GB  ratio {t / c:5.2f}  "
              f"excess {(t - c) / 1e9:6.2f} GB  {sum(r['ms'] for r in rs):7.2f} ms")
    print("\nweights-direct family by shape (N, K), largest excess first:")
    shapes = {}
    for r in rows:
        if r["family"] != "conv_gemm_wd_kernel":
            continue
        s = shapes.setdefault((r["n"], r["k"], r["m"]), [0, 0.0, 0.0, 0.0, 0.0])
        s[0] += 1; s[1] += r["bytes"]; s[2] += r["read"]; s[3] += r["written"]; s[4] += r["ms"]
    for (nn, kk, mm), (cnt, c, rd, wb, ms) in sorted(shapes.items(), key=lambda kv: -(kv[1][2] + kv[1][3] - kv[1][1])):
        wbytes = nn * kk * 4
        a_bytes = c - wbytes - wb  # what the compulsory figure holds for the activations read (approximately: outputs ~ written)
        print(f"  N {nn:5d} K {kk:5d} M {mm:8d} x{cnt:3d}: compulsory {c / 1e6:9.1f} MB, read {rd / 1e6:9.1f}, written {wb / 1e6:8.1f}, ratio "
              f"{(rd + wb) / c:5.2f}, excess {(rd + wb - c) / 1e6:8.1f} MB; weights {wbytes / 1e6 * cnt:7.1f} MB = {wbytes / 1e6:5.1f} each; "
              f"m tiles {-(-mm // 128)}, n tiles {nn // 256}; {ms:6.2f} ms")


if __name__ == "__main__":
    if sys.argv[1] == "log":
        run(sys.argv[2])
    elif sys.argv[1] == "run":
        run()
    else:
        show(*sys.argv[2:5])
