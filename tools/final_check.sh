# What is run on the GPU box before a round is closed (two gpurun calls of <= 20 minutes):
#   gpurun --timeout 1200 -- 'bash tools/final_check.sh tests'      the whole -m gpu suite (-s: every printed margin), smoke(), the
#                                                                    two-rank rehearsal of bench.py --gpus 2 on one GPU (gloo)
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh main'   then `... face`: the measurement set under gpurun_out/refresh
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/final
python -m pytest tests -q -m gpu -s > gpurun_out/final/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/final/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/final/smoke.log
AVCER_BENCH_REHEARSE=1 timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu --no-secondary --no-configs > gpurun_out/final/rehearse2.json 2> gpurun_out/final/rehearse2.err; echo "rehearse rc=$?"; tail -c 400 gpurun_out/final/rehearse2.json
