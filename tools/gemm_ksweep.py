#!/usr/bin/env python3
"""Tuning aid: TFLOP/s of the plain forward GEMM vs K (separates per-tile overhead from the main loop)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import run  # noqa: E402
from hig_amd import _lib  # noqa: E402

for N in (1024, 512):
    for K in (256, 512, 1024, 2048, 8192):
        ms, tf = run(12544, N, K, _lib.XF_NONE, _lib.EPI_BIAS, reps=10)
        print("tile=%s N=%d K=%5d  %.3f ms  %.1f TFLOP/s" % (os.environ.get("HIG_GEMM_TILE", "auto"), N, K, ms, tf))
