#!/usr/bin/env python3
"""Micro-benchmark of the fused GEMM at the denoiser's shapes (tuning aid, not the headline).
HIG_GEMM_TILE=0..3 forces 128x128 / 64x128 / 128x64 / 64x64 tiles."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hig_amd import _lib  # noqa: E402


def run(M, N, K, xf, epi, reps=20, warm=3):
    dev = "cuda"
    X = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    if os.environ.get("ZERO"):   # data-dependent power: zero operands show the clock give-back
        X.zero_()
        W.zero_()
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    res = torch.randn(M, N, device=dev)
    st = torch.stack([X.mean(-1), torch.rsqrt(X.var(-1, unbiased=False) + 1e-5)], -1).contiguous()
    g, be = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    ss = torch.randn(M // 196 + 1, 2 * K, device=dev) * 0.1
    d = _lib.GemmDesc()
    d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc = X.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N
    d.I, d.J, d.R, d.xf, d.epi, d.prec = M, N, K, xf, epi, int(os.environ.get("PREC", 0))
    d.bias, d.res, d.ldr = b.data_ptr(), res.data_ptr(), N
    d.stats, d.gamma, d.beta = st.data_ptr(), g.data_ptr(), be.data_ptr()
    d.ss, d.ss_ld, d.ss_shift_off, d.rows_per_sample = ss.data_ptr(), 2 * K, K, 196
    L = _lib.lib()
    tail = torch.zeros(L.hig_gemm_tail_ws_bytes(), dtype=torch.uint8, device=dev)   # HIG_GEMM_TAIL=0 switches the split tail off

    def launch():
        _lib.check(L.hig_gemm_ws(C.byref(d), tail.data_ptr(), tail.numel(), _lib.stream_ptr()))

    for _ in range(warm):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, 2.0 * M * N * K / ms / 1e9


if __name__ == "__main__":
    M = int(os.environ.get("M", 12544))
    shapes = [("qkv  LN+bias", 1536, 512, _lib.XF_LN, _lib.EPI_BIAS),
              ("sty  mod+res", 512, 512, _lib.XF_LN_MOD_SILU, _lib.EPI_BIAS_RES),
              ("ffn1 gelu", 1024, 512, _lib.XF_NONE, _lib.EPI_BIAS_GELU),
              ("ffn2 bias", 512, 1024, _lib.XF_NONE, _lib.EPI_BIAS)]
    for name, N, K, xf, epi in shapes:
        ms, tf = run(M, N, K, xf, epi)
        print("prec=%s tile=%s M=%d %-14s N=%4d K=%4d  %.3f ms  %.1f TFLOP/s" %
              (os.environ.get("PREC", "0"), os.environ.get("HIG_GEMM_TILE", "auto"), M, name, N, K, ms, tf))
