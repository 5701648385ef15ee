"""K = 1024 LayerNorm fold, producer and consumer against the plain launches (config-5 shapes).  usage: ws16_fold1024_time.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hig_amd import _lib
M, d = 9600, 1024
dev = "cuda"; NS = 4
lib = _lib.lib()
X = [torch.randn(M, d, device=dev).to(torch.bfloat16) for _ in range(NS)]
H = [torch.randn(M, d, device=dev).to(torch.bfloat16) for _ in range(NS)]
stats = torch.zeros(M, d // 128, 2, device=dev); stats[:, :, 1] = 128.0
def run(name, J, epi, st_out, st_in):
    W = (torch.randn(J, d, device=dev) * 0.03).to(torch.bfloat16); b = torch.randn(J, device=dev); cs = W.float().sum(1)
    outs = [torch.empty(M, J, device=dev, dtype=torch.bfloat16) for _ in range(NS)] if epi == _lib.EPI_BIAS else H
    ds = []
    for i in range(NS):
        g = _lib.Gemm16Desc()
        g.X, g.ldx, g.Y, g.ldy, g.C, g.ldc, g.c_f32 = X[i].data_ptr(), d, W.data_ptr(), d, outs[i].data_ptr(), J, 0
        g.I, g.J, g.R, g.epi, g.bias = M, J, d, epi, b.data_ptr()
        if epi == _lib.EPI_BIAS_RES: g.res, g.ldr, g.res_f32 = outs[i].data_ptr(), J, 0
        if st_out: g.row_stats_out = stats.data_ptr()
        if st_in: g.row_stats_in, g.ln_colsum = stats.data_ptr(), cs.data_ptr()
        ds.append(g)
    for i in range(NS): _lib.check(lib.hig_gemm_bf16(C.byref(ds[i]), _lib.stream_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(40): _lib.check(lib.hig_gemm_bf16(C.byref(ds[r % NS]), _lib.stream_ptr()))
    e1.record(); torch.cuda.synchronize()
    print("%-40s %7.1f us" % (name, e0.elapsed_time(e1) / 40 * 1e3))
for rep in range(2):
    run("q/k/v 3072x1024 bias", 3 * d, _lib.EPI_BIAS, False, False)
    run("q/k/v 3072x1024 bias, folded LN", 3 * d, _lib.EPI_BIAS, False, True)
    run("ca_q 1024x1024 bias", d, _lib.EPI_BIAS, False, False)
    run("ca_q 1024x1024 bias, folded LN", d, _lib.EPI_BIAS, False, True)
    run("sty_out 1024x1024 bias+res", d, _lib.EPI_BIAS_RES, False, False)
    run("sty_out 1024x1024 bias+res + stats out", d, _lib.EPI_BIAS_RES, True, False)
