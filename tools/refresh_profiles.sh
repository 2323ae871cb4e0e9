#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the measurement set the docs quote, written under gpurun_out/ (copy into profiles/ after).
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh'
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/refresh
rm -rf "$O" && mkdir -p "$O"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o x3 -- python3 bench.py --steps 5 --warmup 2 --no-secondary --no-cpu --no-configs --no-overlap > "$O/stats.log" 2>&1
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_FETCH_SIZE" -- python3 bench.py --steps 1 --warmup 1 --no-secondary --no-cpu --no-configs --no-events --no-overlap > "$O/pmc_f.log" 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_WRITE_SIZE" -- python3 bench.py --steps 1 --warmup 1 --no-secondary --no-cpu --no-configs --no-events --no-overlap > "$O/pmc_w.log" 2>&1
echo "write done"
python3 tools/pmc_traffic.py "$O/pmc_FETCH_SIZE" "$O/pmc_WRITE_SIZE" --out "$O/traffic_x3.json" > "$O/traffic_x3.txt"
find "$O/stats" -name "*kernel_stats.csv" -exec cp {} "$O/kernel_stats.csv" \;
# the raw counter dumps are large: keep the summaries only
rm -rf "$O/pmc_FETCH_SIZE" "$O/pmc_WRITE_SIZE" "$O/stats"
# the bench line quotes the PMC traffic of THESE sources: refresh the committed figure first, then take the line
cp "$O/traffic_x3.json" profiles/r05_traffic_x3.json
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > "$O/bench_line.json" 2> "$O/bench.err"
echo "bench done" && tail -c 300 "$O/bench_line.json"
timeout -k 10 300 python3 tools/trace_step.py run > /dev/null 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -o t -- python3 tools/trace_step.py run > "$O/trace.log" 2>&1
python3 tools/trace_step.py show "$O/trace" > "$O/step_trace.txt" 2>&1 || true
rm -rf "$O/trace"
# one frame / one window per call (the drop-in mirrors) and BASELINE config 2, launch by launch
for w in static audio lstm; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$O/ts_$w" -o t -- python3 tools/trace_static.py run 1 $w > "$O/ts_$w.log" 2>&1
  m=stem_pool; [ $w = audio ] && m=wav_normalize; [ $w = lstm ] && m=gather_or_first
  python3 tools/trace_static.py show "$O/ts_$w" $m > "$O/per_call_trace_$w.txt" 2>&1 || true
  rm -rf "$O/ts_$w"
done
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$O/ts_b256" -o t -- python3 tools/trace_static.py run 256 > "$O/ts_b256.log" 2>&1
python3 tools/trace_static.py show "$O/ts_b256" > "$O/static_b256_trace.txt" 2>&1 || true
rm -rf "$O/ts_b256"
timeout -k 10 200 python3 tools/clock_probe.py 3 > "$O/clock_probe.txt" 2>&1
timeout -k 10 300 python3 tools/ab_layers.py --frames 1024 > "$O/ab_layers.txt" 2>&1
timeout -k 10 300 python3 tools/ab_layers.py --frames 2048 --only "l3." > "$O/ab_layers_2048_l3.txt" 2>&1
timeout -k 10 300 python3 tools/ab_layers.py --frames 2048 --only "l4." > "$O/ab_layers_2048_l4.txt" 2>&1
ls -la "$O"
