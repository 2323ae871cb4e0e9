#!/usr/bin/env python3
"""Launch-by-launch listing of one detector pass (the last one of `tools/face_run.py`) from a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 tools/face_run.py 1
    python3 tools/face_trace.py out
"""
import csv
import glob
import re
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# every sub-pass of the detector begins with its stem: the fused stem_pool_u8_kernel<true> in the x3 mode, face_pre_kernel otherwise
starts = [i for i, r in enumerate(rows) if "stem_pool_u8_kernel" in r["Kernel_Name"] or "face_pre_kernel" in r["Kernel_Name"]]
# a 750-frame pass is three sub-passes (273 + 273 + 204 frames): the last pass begins at the third such launch from the end
first = starts[-3] if len(starts) >= 3 else starts[0]
t0 = int(rows[first]["Start_Timestamp"])
total, by = 0.0, {}
for r in rows[first:]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    total += dur
    by[n] = by.get(n, 0.0) + dur
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} us  {dur:9.1f} us  grid {r.get('Grid_Size', '?'):>9s}  {n[:110]}")
print(f"sum of kernel durations in the pass: {total / 1e3:.2f} ms; span {(int(rows[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms")
for n, v in sorted(by.items(), key=lambda kv: -kv[1]):
    print(f"{v / 1e3:9.2f} ms  {n[:120]}")
