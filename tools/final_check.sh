set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests -x -q -m gpu -s > gpurun_out/r5z_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r5z_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5z_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5z_smoke.log
AVCER_BENCH_REHEARSE=1 timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu --no-secondary --no-configs > gpurun_out/r5z_rehearse2.json 2> gpurun_out/r5z_rehearse2.err; echo "rehearse rc=$?"; tail -c 400 gpurun_out/r5z_rehearse2.json
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1; echo "refresh rc=$?"
