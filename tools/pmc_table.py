#!/usr/bin/env python3
"""Per-kernel HBM traffic of ANY profiled program from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; gfx950 corrections as
in tools/pmc_traffic.py: FETCH_SIZE x 2, KiB units):   python3 tools/pmc_table.py <fetch dir> <write dir> [rows=30]"""
import sys

from pmc_traffic import per_kernel

fe, wr = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
rows = []
for name, (n, kib, secs) in fe.items():
    wn, wkib, wsecs = wr.get(name, [0, 0.0, 0.0])
    rows.append((2.0 * kib * 1024.0 + wkib * 1024.0, name, n, 2.0 * kib * 1024.0, wkib * 1024.0, secs))
rows.sort(reverse=True)
print(f"{'kernel':90s} {'launches':>8s} {'read GB':>9s} {'write GB':>9s} {'ms':>9s} {'TB/s':>6s}")
for tot, name, n, rd, wb, secs in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{name[-90:]:90s} {n:8d} {rd / 1e9:9.2f} {wb / 1e9:9.2f} {secs * 1e3:9.2f} {tot / secs / 1e12 if secs else 0:6.2f}")
print(f"total: {sum(r[3] for r in rows) / 1e9:.1f} GB read, {sum(r[4] for r in rows) / 1e9:.1f} GB written, {sum(r[5] for r in rows) * 1e3:.1f} ms of kernels (under the counters)")
