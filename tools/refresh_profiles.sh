#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the measurement set the docs quote, written under gpurun_out/ (copy into profiles/ after).
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh'
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/refresh
rm -rf "$O" && mkdir -p "$O"
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > "$O/bench_line.json" 2> "$O/bench.err"
echo "bench done" && tail -c 300 "$O/bench_line.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o x3 -- python3 bench.py --steps 5 --warmup 2 --no-secondary --no-cpu --no-configs > "$O/stats.log" 2>&1
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_FETCH_SIZE" -- python3 bench.py --steps 1 --warmup 1 --no-secondary --no-cpu --no-configs > "$O/pmc_f.log" 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_WRITE_SIZE" -- python3 bench.py --steps 1 --warmup 1 --no-secondary --no-cpu --no-configs > "$O/pmc_w.log" 2>&1
echo "write done"
python3 tools/pmc_traffic.py "$O/pmc_FETCH_SIZE" "$O/pmc_WRITE_SIZE" --out "$O/traffic_x3.json" > "$O/traffic_x3.txt"
find "$O/stats" -name "*kernel_stats.csv" -exec cp {} "$O/kernel_stats.csv" \;
# the raw counter dumps are large: keep the summaries only
rm -rf "$O/pmc_FETCH_SIZE" "$O/pmc_WRITE_SIZE" "$O/stats"
timeout -k 10 300 python3 tools/layer_bench.py --dtype x3s --frames 1024 --chunks 128 --iters 5 > "$O/layers_x3.txt" 2>&1
ls -la "$O"
