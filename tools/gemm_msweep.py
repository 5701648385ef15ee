#!/usr/bin/env python3
"""Tuning aid: fp32 forward GEMM (N=1024, K=512) vs M -- exposes tile-count quantisation / tail effects."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import run  # noqa: E402
from hig_amd import _lib  # noqa: E402

for M in (4096, 8192, 12288, 12544, 13312, 16384, 32768):
    for K in (512, 2048):
        ms, tf = run(M, 1024, K, _lib.XF_NONE, _lib.EPI_BIAS, reps=10)
        print("tile=%s M=%5d N=1024 K=%4d tiles64=%5d  %.3f ms  %.1f TFLOP/s" %
              (os.environ.get("HIG_GEMM_TILE", "auto"), M, K, (M // 64) * 16, ms, tf))
