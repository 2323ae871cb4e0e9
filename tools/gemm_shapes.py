#!/usr/bin/env python3
"""Quick A/B harness: time a handful of representative conv_gemm shapes (GPU). Usage: gemm_shapes.py [f32|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.layer_bench import conv2d, linear, run  # noqa: E402
from avcer_amd.engine import Engine  # noqa: E402

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    eng = Engine(0)
    layers = [
        linear(12672, 1024, 3072, "qkv 1024->3072"),
        linear(12672, 4096, 1024, "ffn2 4096->1024", res=True),
        conv2d(256, 14, 256, 3, 1, 1, 256, "l3 c2 3x3 256"),
        conv2d(256, 55, 64, 1, 1, 0, 256, "l1 c3 64->256+res", res=True),
        conv2d(256, 55, 256, 1, 1, 0, 64, "l1 c1 256->64"),
        conv2d(256, 28, 128, 3, 1, 1, 128, "l2 c2 3x3 128"),
    ]
    run(eng, layers, {"f32": 0, "bf16": 1, "x3": 3, "x3s": 5}[which], 10, which)
