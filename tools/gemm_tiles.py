#!/usr/bin/env python3
"""Per-shape tile comparison for the forward GEMMs of config 2 (plain operands, the path's epilogues).
Run once per HIG_GEMM_TILE value (the knob is read once per process): prints one line per shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import run, _lib  # noqa: E402

M = int(os.environ.get("M", 12544))
for name, N, K, epi in (("qkv bias", 1536, 512, _lib.EPI_BIAS), ("proj bias+res", 512, 512, _lib.EPI_BIAS_RES),
                        ("ffn1 gelu", 1024, 512, _lib.EPI_BIAS_GELU), ("ffn2 bias+res", 512, 1024, _lib.EPI_BIAS_RES)):
    ms, tf = run(M, N, K, _lib.XF_NONE, epi, reps=40, warm=5)
    print("tile=%s percu=%s %-14s N=%4d K=%4d  %.1f us  %.1f TFLOP/s" %
          (os.environ.get("HIG_GEMM_TILE", "auto"), os.environ.get("HIG_GEMM_PERCU", "-"), name, N, K, ms * 1e3, tf))
