#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the measurement set the docs quote, written under gpurun_out/ (copy into profiles/ after).
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh'
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/refresh
PART=${1:-all}   # main | face | all (two gpurun calls of <= 20 minutes: `bash tools/refresh_profiles.sh main`, then `... face`)
mkdir -p "$O"
if [ "$PART" != face ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o x3 -- python3 bench.py --steps 5 --warmup 2 --no-secondary --no-cpu --no-configs --no-overlap --one-lane > "$O/stats.log" 2>&1
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_FETCH_SIZE" -- python3 bench.py --steps 1 --warmup 1 --no-secondary --no-cpu --no-configs --no-events --no-overlap --one-lane > "$O/pmc_f.log" 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_WRITE_SIZE" -- python3 bench.py --steps 1 --warmup 1 --no-secondary --no-cpu --no-configs --no-events --no-overlap --one-lane > "$O/pmc_w.log" 2>&1
echo "write done"
python3 tools/pmc_traffic.py "$O/pmc_FETCH_SIZE" "$O/pmc_WRITE_SIZE" --out "$O/traffic_x3.json" > "$O/traffic_x3.txt"
find "$O/stats" -name "*kernel_stats.csv" -exec cp {} "$O/kernel_stats.csv" \;
# the raw counter dumps are large: keep the summaries only
rm -rf "$O/pmc_FETCH_SIZE" "$O/pmc_WRITE_SIZE" "$O/stats"
# the bench line quotes the PMC traffic of THESE sources: refresh the committed figure first, then take the line
cp "$O/traffic_x3.json" profiles/r06_traffic_x3.json
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
fi
if [ "$PART" = main ]; then ls -la "$O"; exit 0; fi
# stage 0 (detector) evidence: kernel stats, HBM traffic, launch-by-launch trace, families
{ timeout -k 10 200 python3 tools/face_run.py 3 2>&1; timeout -k 10 200 python3 tools/face_run.py 3 fam 2>&1; } | grep -v amdgpu.ids > "$O/face_families.txt" || true
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/face_stats" -o face -- python3 tools/face_run.py 3 serial > "$O/face_stats.log" 2>&1
find "$O/face_stats" -name "*kernel_stats.csv" -exec cp {} "$O/face_kernel_stats.csv" \;
rm -rf "$O/face_stats"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_ff" -- python3 tools/face_run.py 1 serial > "$O/pmc_ff.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_fw" -- python3 tools/face_run.py 1 serial > "$O/pmc_fw.log" 2>&1
(cd tools && python3 pmc_table.py "../$O/pmc_ff" "../$O/pmc_fw" 40) > "$O/face_traffic.txt" 2>&1 || true
rm -rf "$O/pmc_ff" "$O/pmc_fw"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/ftr" -o t -- python3 tools/face_run.py 1 serial > "$O/ftr.log" 2>&1
python3 tools/face_trace.py "$O/ftr" > "$O/face_trace.txt" 2>&1 || true
rm -rf "$O/ftr"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/ri" -o t -- python3 tools/trace_run_inference.py run > "$O/ri.log" 2>&1
python3 tools/trace_run_inference.py show "$O/ri" > "$O/run_inference_trace.txt" 2>&1 || true
rm -rf "$O/ri"
timeout -k 10 200 python3 tools/x3_headroom.py 2>&1 | grep -v amdgpu.ids > "$O/x3_headroom.txt" || true
ls -la "$O"
