"""Stylization-out GEMM (M x 512 x 512, bias + residual, in place) with and without the row statistics of the LayerNorm
fold.  usage: ws16_stats_time.py [B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hig_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M, d = B * 196, 512
dev = "cuda"
NS = 4
X = [torch.randn(M, d, device=dev).to(torch.bfloat16) for _ in range(NS)]
H = [torch.randn(M, d, device=dev).to(torch.bfloat16) for _ in range(NS)]
W = (torch.randn(d, d, device=dev) * 0.05).to(torch.bfloat16); b = torch.randn(d, device=dev)
stats = torch.empty(M, 4, 2, device=dev)
lib = _lib.lib()
def desc(i, with_stats):
    g = _lib.Gemm16Desc()
    g.X, g.ldx, g.Y, g.ldy, g.C, g.ldc, g.c_f32 = X[i].data_ptr(), d, W.data_ptr(), d, H[i].data_ptr(), d, 0
    g.I, g.J, g.R, g.epi, g.bias = M, d, d, _lib.EPI_BIAS_RES, b.data_ptr()
    g.res, g.ldr, g.res_f32 = H[i].data_ptr(), d, 0
    if with_stats: g.row_stats_out = stats.data_ptr()
    return g
for ws in (0, 1, 0, 1):
    ds = [desc(i, ws) for i in range(NS)]
    for i in range(NS): _lib.check(lib.hig_gemm_bf16(C.byref(ds[i]), _lib.stream_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(40): _lib.check(lib.hig_gemm_bf16(C.byref(ds[r % NS]), _lib.stream_ptr()))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 40 * 1e3
    print("B=%d M=%d stats=%d: %.1f us  %.0f TFLOP/s" % (B, M, ws, us, 2.0 * M * d * d / us / 1e6))
