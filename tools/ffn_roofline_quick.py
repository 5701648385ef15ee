#!/usr/bin/env python3
"""bench.py's FFN-linear1 roofline microbenchmark alone (8 operand sets = 617 MB: activations from HBM).  HIG_LIB_ALT: variant build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hig_amd import _lib
if os.environ.get("HIG_LIB_ALT"): _lib.LIB_PATH = os.environ["HIG_LIB_ALT"]
import bench
r = bench.ffn_gemm_roofline(dict(bench.CFG), torch.device("cuda", 0))
print("ffn1 HBM-resident: %.4f ms  %.1f TFLOP/s  frac %.4f   batches %s" % (r["avg_launch_ms"], r["achieved"], r["frac"], r["batch_avg_launch_ms"]))
