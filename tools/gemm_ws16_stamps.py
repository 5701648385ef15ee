"""Where a workgroup of the weight-stationary bf16 GEMM (csrc/gemm_ws16.hip) spends its time: s_memtime stamps.
usage: python tools/gemm_ws16_stamps.py [shape] [B]   (HIG_BF16_WS_NWJ / HIG_BF16_WS_SLOTS select the variant)"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hig_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "ffn1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
M = B * 196
I, J, R, epi = {"ffn1": (M, 1024, 512, _lib.EPI_BIAS_GELU), "qkv": (M, 1536, 512, _lib.EPI_BIAS), "ca_q": (M, 512, 512, _lib.EPI_BIAS),
                "sty_out": (M, 512, 512, _lib.EPI_BIAS_RES), "ffn2": (M, 512, 1024, _lib.EPI_BIAS)}[name]
dev = "cuda"
X = torch.randn(I, R, device=dev).to(torch.bfloat16); W = (torch.randn(J, R, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(J, device=dev); out = torch.empty(I, J, device=dev, dtype=torch.bfloat16); res = torch.randn(I, J, device=dev).to(torch.bfloat16)
d = _lib.Gemm16Desc()
d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc, d.c_f32 = X.data_ptr(), R, W.data_ptr(), R, out.data_ptr(), J, 0
d.I, d.J, d.R, d.epi, d.bias = I, J, R, epi, b.data_ptr()
if epi == _lib.EPI_BIAS_RES: d.res, d.ldr, d.res_f32 = res.data_ptr(), J, 0
lib = _lib.lib()
NB = 1024
stamps = torch.zeros(NB * 16, dtype=torch.int64, device=dev)
lib.hig_gemm_ws16_debug_stamps(C.c_void_p(stamps.data_ptr()))
junk = torch.ones(256 << 20, device=dev)
for it in range(3):
    junk.sum().item(); stamps.zero_()
    _lib.check(lib.hig_gemm_bf16(C.byref(d), _lib.stream_ptr())); torch.cuda.synchronize()
    s = stamps.view(NB, 16).cpu()
    s = s[s[:, 0] > 0]
    nt = s[:, 14]
    print("%s B=%d run %d: %d workgroups, tiles per workgroup %d..%d" % (name, B, it, len(s), nt.min(), nt.max()))
    def show(label, a, b_, sel=None):
        dlt = (s[:, b_] - s[:, a]).double()
        if sel is not None: dlt = dlt[sel]
        if len(dlt): print("   %-44s median %7.0f cycles   p10 %7.0f   p90 %7.0f   (n=%d)" % (label, dlt.median(), dlt.quantile(0.1), dlt.quantile(0.9), len(dlt)))
    show("requests issued (W loads, X(0) DMA, bias)", 0, 1)
    show("W + X(0) landed", 1, 2)
    for t in range(int(min(nt.max(), 8))):
        show("iteration %d (barrier to barrier)" % t, 3 + t, 4 + t if t + 1 < 9 else 12, nt > t + 1)
    full = nt == nt.max()
    lastcol = 3 + int(nt.max()) - 1
    show("last iteration -> drain barrier", lastcol, 12, full)
    show("drain (last epilogue + stores)", 12, 13)
    show("whole workgroup", 0, 13)
lib.hig_gemm_ws16_debug_stamps(None)
