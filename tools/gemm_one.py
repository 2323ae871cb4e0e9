#!/usr/bin/env python3
"""One conv_gemm shape, a few launches: the target of `rocprofv3 --pmc ...` counter passes.

    python3 tools/gemm_one.py l3c2|qkv|ffn2|l1c3 [x3s|bf16|f32] [iters]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.layer_bench import conv2d, linear, run  # noqa: E402
from avcer_amd.engine import Engine  # noqa: E402

SHAPES = {
    "l3c2": lambda: conv2d(1024, 14, 256, 3, 1, 1, 256, "l3 c2 3x3 256"),
    "l2c2": lambda: conv2d(1024, 28, 128, 3, 1, 1, 128, "l2 c2 3x3 128"),
    "qkv": lambda: linear(12672, 1024, 3072, "qkv 1024->3072"),
    "ffn2": lambda: linear(12672, 4096, 1024, "ffn2 4096->1024", res=True),
    "l1c3": lambda: conv2d(1024, 55, 64, 1, 1, 0, 256, "l1 c3 64->256+res", res=True),
    "big": lambda: linear(131072, 2048, 2048, "square 128k x 2048 x 2048"),
}

if __name__ == "__main__":
    shape = sys.argv[1] if len(sys.argv) > 1 else "l3c2"
    which = sys.argv[2] if len(sys.argv) > 2 else "x3s"
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    run(Engine(0), [SHAPES[shape]()], {"f32": 0, "bf16": 1, "x3": 3, "x3s": 5, "x3w": 7}[which], iters, f"{shape} {which}")
