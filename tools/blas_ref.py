#!/usr/bin/env python3
"""Reference point (not used by the product): what the vendor fp32 GEMM (torch.mm -> rocBLAS / hipBLASLt)
reaches on the denoiser's shapes, next to our kernel (tools/gemm_bench.py)."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"
for (M, N, K) in ((12544, 1024, 512), (12544, 512, 1024), (12544, 1536, 512), (12544, 512, 512), (16384, 1024, 8192)):
    X = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    out = torch.empty(M, N, device=dev)
    for _ in range(3):
        torch.mm(X, W.t(), out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        torch.mm(X, W.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("torch.mm fp32 M=%d N=%d K=%d  %.3f ms  %.1f TFLOP/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
