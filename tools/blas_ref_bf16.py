#!/usr/bin/env python3
"""Reference point (not used by the product): what the vendor bf16 GEMM (torch.mm / F.linear -> hipBLASLt) reaches on the
bf16-storage denoiser's shapes, operands rotating over enough sets to defeat the Infinity Cache, next to our kernels
(tools/gemm16_shapes.py).  The product never calls these."""
import torch
import torch.nn.functional as F

dev = "cuda"
SHAPES = (("ffn1", 12544, 1024, 512), ("ffn2", 12544, 512, 1024), ("qkv", 12544, 1536, 512), ("sty_out", 12544, 512, 512),
          ("cfg5 qkv", 9600, 3072, 1024), ("cfg5 ffn1", 9600, 1024, 1024), ("big", 16384, 8192, 8192))
for name, M, N, K in SHAPES:
    nset = max(2, min(12, int(600e6 // ((M * K + M * N) * 2))))
    Xs = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nset)]
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
    for mode in ("mm", "linear+bias"):
        def run(i):
            if mode == "mm":
                torch.mm(Xs[i % nset], W.t(), out=outs[i % nset])
            else:
                outs[i % nset] = F.linear(Xs[i % nset], W, b)
        for i in range(6):
            run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        reps = 48
        e0.record()
        for i in range(reps):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%-10s %-12s M=%d N=%d K=%d  %7.1f us  %7.1f TFLOP/s = %.3f of 2500" %
              (name, mode, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 2500))
