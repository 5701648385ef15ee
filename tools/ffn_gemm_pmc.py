#!/usr/bin/env python3
"""Launches the FFN linear1 GEMM (M=12544, K=512, N=1024, bias+GELU) a few times so that
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) can price its HBM traffic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tools.gemm_bench import run  # noqa: E402
from hig_amd import _lib  # noqa: E402

def warm(n=400, nb=8):
    """`warm` mode: n launches back to back rotating over nb operand sets (8 x 77 MB > the Infinity Cache: HBM-resident
    activations, as in bench.py's roofline microbenchmark and inside the forward) -- the counters of the kernel in its steady
    regime, next to the cold-cache launches of the default mode.  400 launches: a fresh process finds the chip idle, and the first
    hundred or so launches run 15 % slower than the steady state (124 us against 106 us by kernel trace at a HIGHER shader clock:
    the memory side has not ramped up yet) -- summarise the LAST 100."""
    import ctypes as C
    M, K, N = 12544, 512, 1024
    L = _lib.lib()
    W, b = torch.randn(N, K, device="cuda") * 0.05, torch.randn(N, device="cuda")
    sets = []
    for _ in range(nb):
        X, out = torch.randn(M, K, device="cuda"), torch.empty(M, N, device="cuda")
        d = _lib.GemmDesc()
        d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc, d.I, d.J, d.R = X.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, M, N, K
        d.epi, d.bias = _lib.EPI_BIAS_GELU, b.data_ptr()
        sets.append((d, X, out))
    tail = torch.zeros(L.hig_gemm_tail_ws_bytes(), dtype=torch.uint8, device="cuda")
    for i in range(n):
        _lib.check(L.hig_gemm_ws(C.byref(sets[i % nb][0]), tail.data_ptr(), tail.numel(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    print("ffn1 warm: %d launches" % n)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "warm":
        warm()
        sys.exit(0)
    # evict with a READ sweep of 1 GiB: the caches end up full of CLEAN junk lines, so the GEMM reads from
    # HBM (not the 256 MiB MALL) and no write-back of somebody else's dirty lines is billed to it
    junk = torch.ones(256 << 20, device="cuda")
    torch.cuda.synchronize()
    for _ in range(6):     # every profiled launch starts cold: one GEMM per sweep, no warm-up launches
        junk.sum().item()
        ms, tf = run(12544, 1024, 512, _lib.XF_NONE, _lib.EPI_BIAS_GELU, reps=1, warm=0)
    print("ffn1 %.3f ms %.1f TF" % (ms, tf))
