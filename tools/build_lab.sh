#!/bin/bash
# Lab build (NOT the product): libavcer_hip.so once more with the round-3 bf16 operand split (-DAVCER_SPLIT_BF16, see
# csrc/split_dev.h), for same-box A/B runs of the split type (tools/ab_split.py).  Output: tools/lab/libavcer_hip_bf16split.so
# Sources and flags come from avcer_amd/build.py (SOURCES, FLAGS): the two builds differ in the one -D switch and nothing else.
set -e
cd "$(dirname "$0")/.."
python3 -c 'from avcer_amd import build; print(build.build(force=True, extra_flags=["-DAVCER_SPLIT_BF16=1"], out="tools/lab/libavcer_hip_bf16split.so", tag=".bf16split"))'
ls -la tools/lab/libavcer_hip_bf16split.so
