#!/bin/bash
# Lab build (NOT the product): libavcer_hip.so once more with the round-3 bf16 operand split (-DAVCER_SPLIT_BF16, see
# csrc/split_dev.h), for same-box A/B runs of the split type (tools/ab_split.py).  Output: tools/lab/libavcer_hip_bf16split.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/lab
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DAVCER_SPLIT_BF16=1"
for f in gemm fused kernels api; do
    hipcc $FLAGS -c avcer_amd/csrc/$f.hip -o tools/lab/$f.bf16split.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC tools/lab/*.bf16split.o -o tools/lab/libavcer_hip_bf16split.so
ls -la tools/lab/libavcer_hip_bf16split.so
