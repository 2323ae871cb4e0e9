#!/bin/bash
# round 6, first GPU call: tests, front-pass sweep, detector evidence, headroom report
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6a
mkdir -p "$O"
timeout -k 10 600 python3 -m pytest tests -m gpu -q > "$O/pytest.txt" 2>&1 || { tail -30 "$O/pytest.txt"; grep -q "passed" "$O/pytest.txt" || exit 1; }
tail -3 "$O/pytest.txt"
timeout -k 10 300 python3 tools/front_batch_sweep.py > "$O/front_batch_sweep.txt" 2>&1
cat "$O/front_batch_sweep.txt"
timeout -k 10 200 python3 tools/face_run.py 3 fam > "$O/face_families.txt" 2>&1
cat "$O/face_families.txt"
timeout -k 10 200 python3 tools/x3_headroom.py > "$O/x3_headroom.txt" 2>&1
tail -5 "$O/x3_headroom.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/face_stats" -o face -- python3 tools/face_run.py 3 > "$O/face_stats.log" 2>&1
find "$O/face_stats" -name "*kernel_stats.csv" -exec cp {} "$O/face_kernel_stats.csv" \;
rm -rf "$O/face_stats"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_f" -- python3 tools/face_run.py 1 > "$O/pmc_f.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_w" -- python3 tools/face_run.py 1 > "$O/pmc_w.log" 2>&1
(cd tools && python3 pmc_table.py "../$O/pmc_f" "../$O/pmc_w" 40) > "$O/face_traffic.txt" 2>&1
rm -rf "$O/pmc_f" "$O/pmc_w"
head -30 "$O/face_kernel_stats.csv"
cat "$O/face_traffic.txt"
