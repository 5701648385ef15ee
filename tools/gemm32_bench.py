#!/usr/bin/env python3
"""Exact-fp32 GEMMs of the config-2 forward, one launch shape per line: time per launch (HIP events over `reps`
back-to-back launches, operand sets rotating so that nothing is served from a warm L2 by accident), TFLOP/s, fraction of
the 157.3 TFLOP/s fp32 matrix peak.  Run it twice in one gpurun call -- HIG_F32_WSP=0 (the tiled kernel of gemm.hip) and
HIG_F32_WSP=1 (gemm_wsp32.hip) -- for the A/B on one box.  M=<rows> overrides 12544."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hig_amd import _lib  # noqa: E402
if os.environ.get("HIG_LIB_ALT"): _lib.LIB_PATH = os.environ["HIG_LIB_ALT"]   # (A/B of a variant build)

PEAK = 157.3


def run(M, N, K, epi, fold=0, reps=30, warm=5, nset=3):
    dev = "cuda"
    L = _lib.lib()
    sets = []
    for s in range(nset):
        X = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        res = torch.randn(M, N, device=dev)
        d = _lib.GemmDesc()
        d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc = X.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N
        d.I, d.J, d.R, d.epi = M, N, K, epi
        d.bias, d.res, d.ldr = b.data_ptr(), res.data_ptr(), N
        keep = [X, W, b, out, res]
        if fold == 1:
            st = torch.empty(M, N // 64, 2, device=dev)
            d.row_stats_out = st.data_ptr()
            keep.append(st)
        if fold == 2:
            xp = X.view(M, K // 64, 64)
            st = torch.stack([xp.sum(-1), ((xp - xp.mean(-1, keepdim=True)) ** 2).sum(-1)], -1).contiguous()
            cs = W.sum(-1).contiguous()
            d.row_stats_in, d.ln_colsum = st.data_ptr(), cs.data_ptr()
            keep += [st, cs]
        sets.append((d, keep))
    tail = torch.zeros(L.hig_gemm_tail_ws_bytes(), dtype=torch.uint8, device=dev)

    def launch(i):
        _lib.check(L.hig_gemm_ws(C.byref(sets[i % nset][0]), tail.data_ptr(), tail.numel(), _lib.stream_ptr()))

    for i in range(warm):
        launch(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0.record()
        for i in range(reps):
            launch(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best, 2.0 * M * N * K / best / 1e9


if __name__ == "__main__":
    M = int(os.environ.get("M", 12544))
    shapes = [("qkv (LN fold in)", 1536, 512, _lib.EPI_BIAS, 2),
              ("qkv bias", 1536, 512, _lib.EPI_BIAS, 0),
              ("sty-out res+stats", 512, 512, _lib.EPI_BIAS_RES, 1),
              ("sty-out res", 512, 512, _lib.EPI_BIAS_RES, 0),
              ("ca-q (LN fold in)", 512, 512, _lib.EPI_BIAS, 2),
              ("ffn1 gelu", 1024, 512, _lib.EPI_BIAS_GELU, 0),
              ("ffn2 bias", 512, 1024, _lib.EPI_BIAS, 0)]
    tot = 0.0
    for name, N, K, epi, fold in shapes:
        ms, tf = run(M, N, K, epi, fold)
        tot += ms
        print("WSP=%s M=%d %-18s N=%4d K=%4d  %7.1f us  %6.1f TFLOP/s  frac %.3f" %
              (os.environ.get("HIG_F32_WSP", "1"), M, name, N, K, ms * 1e3, tf, tf / PEAK), flush=True)
    print("WSP=%s M=%d sum of the seven %.1f us" % (os.environ.get("HIG_F32_WSP", "1"), M, tot * 1e3))
    if M == 12544:   # the text side's key/value projection over the 64 x 77 text rows: one layer, and all eight stacked (round 6)
        for name, N in (("text k/v, one layer", 1024), ("text k/v, 8 layers", 8192)):
            ms, tf = run(4928, N, 256, _lib.EPI_BIAS, 0)
            print("WSP=%s M=%d %-18s N=%4d K=%4d  %7.1f us  %6.1f TFLOP/s  frac %.3f" %
                  (os.environ.get("HIG_F32_WSP", "1"), 4928, name, N, 256, ms * 1e3, tf, tf / PEAK), flush=True)


def run_wgrad(M, I, J, reps=20, warm=3):
    """The weight-gradient form (hig_gemm_split, library's own split count): dW = dC^T . act over M rows + dbias."""
    dev = "cuda"
    L = _lib.lib()
    dC, act = torch.randn(M, I, device=dev), torch.randn(M, J, device=dev)
    out, xs = torch.empty(I, J, device=dev), torch.empty(I, device=dev)
    d = _lib.GemmDesc()
    d.X, d.ldx, d.x_rs, d.Y, d.ldy, d.y_rs = dC.data_ptr(), I, 1, act.data_ptr(), J, 1
    d.C, d.ldc, d.I, d.J, d.R, d.xcolsum = out.data_ptr(), J, I, J, M, xs.data_ptr()
    n = 64 << 20
    slabs = torch.empty(n, device=dev)
    for _ in range(warm):
        _lib.check(L.hig_gemm_split(C.byref(d), 0, slabs.data_ptr(), n, _lib.stream_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            _lib.check(L.hig_gemm_split(C.byref(d), 0, slabs.data_ptr(), n, _lib.stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best, 2.0 * M * I * J / best / 1e9


if __name__ == "__main__" and os.environ.get("WGRAD", "1") != "0":
    M = int(os.environ.get("M", 12544))
    for name, I, J in [("sty-out / ca-q", 512, 512), ("q/k/v", 1536, 512), ("ffn1", 1024, 512), ("ffn2", 512, 1024)]:
        ms, tf = run_wgrad(M, I, J)
        print("WSP=%s M=%d wgrad %-15s dW %4d x %4d  %7.1f us (kernel + slab reduction)  %6.1f TFLOP/s  frac %.3f" %
              (os.environ.get("HIG_F32_WSP", "1"), M, name, I, J, ms * 1e3, tf, tf / PEAK), flush=True)
