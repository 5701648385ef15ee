#!/usr/bin/env python3
"""Launches the FFN linear1 GEMM (M=12544, K=512, N=1024, bias+GELU) a few times so that
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) can price its HBM traffic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tools.gemm_bench import run  # noqa: E402
from hig_amd import _lib  # noqa: E402

if __name__ == "__main__":
    # evict with a READ sweep of 1 GiB: the caches end up full of CLEAN junk lines, so the GEMM reads from
    # HBM (not the 256 MiB MALL) and no write-back of somebody else's dirty lines is billed to it
    junk = torch.ones(256 << 20, device="cuda")
    torch.cuda.synchronize()
    for _ in range(6):     # every profiled launch starts cold: one GEMM per sweep, no warm-up launches
        junk.sum().item()
        ms, tf = run(12544, 1024, 512, _lib.XF_NONE, _lib.EPI_BIAS_GELU, reps=1, warm=0)
    print("ffn1 %.3f ms %.1f TF" % (ms, tf))
