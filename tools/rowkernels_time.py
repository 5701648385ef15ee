#!/usr/bin/env python3
"""The row kernels of the fp32 forward alone on the chip (bench.hbm_kernel_rooflines): us per launch, GB/s of algorithmic bytes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hig_amd import _lib
if os.environ.get("HIG_LIB_ALT"): _lib.LIB_PATH = os.environ["HIG_LIB_ALT"]
c = dict(bench.CFG); c["B"] = int(os.environ.get("B", 64))
for k, v in bench.hbm_kernel_rooflines(c, torch.device("cuda", 0)).items():
    print("%-34s %6.1f us  %6.0f GB/s  (%s)" % (k, v["us"], v["GB_per_s"], v["batch_us"]))
