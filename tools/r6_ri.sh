#!/bin/bash
# run_inference with and without the detector (bench legs only) + detector families
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-r6e}
mkdir -p "$O"
timeout -k 10 300 python3 -m pytest tests/test_gpu_face.py tests/test_gpu_retina.py tests/test_gpu_run.py tests/test_gpu_dropin.py -q -x > "$O/pytest.txt" 2>&1; tail -3 "$O/pytest.txt"
timeout -k 10 200 python3 tools/face_run.py 3 fam 2>&1 | grep -v amdgpu.ids | tee "$O/face_families.txt"
timeout -k 10 400 python3 bench.py --steps 3 --warmup 2 --no-secondary --no-cpu > "$O/bench_ri.json" 2> "$O/bench_ri.err"
python3 - "$O/bench_ri.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
c = d["configs"]
r = c["run_inference"]
print("static_b256", {k: (round(v["ms"], 3), round(v.get("ms_one_lane", 0), 3)) for k, v in c["static_b256"].items() if isinstance(v, dict)})
print("run_inference x3 %.1f ms, fp32 %.1f ms; with detector %.1f ms; detector alone %.1f ms (mfma %.1f)" % (
    r["modes"]["x3"]["s"] * 1e3, r["modes"]["fp32"]["s"] * 1e3, r["with_detector"]["s"] * 1e3,
    r["with_detector"]["detector_alone"]["ms"], r["with_detector"]["detector_alone"]["mfma_kernel_ms"]))
PY
