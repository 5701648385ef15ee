"""Where a workgroup of the specialised-wave weight-stationary bf16 GEMM (csrc/gemm_wsp16.hip) spends its time: s_memtime
stamps of thread 0 (a matrix wave).  usage: python tools/gemm_wsp16_stamps.py [shape] [B] [hot]
`hot`: the same operands every run, no cache flush in between (X served by the L2 / Infinity Cache instead of HBM)."""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hig_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "ffn1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
hot = len(sys.argv) > 3 and sys.argv[3] == "hot"
M = B * 196
I, J, R, epi = {"ffn1": (M, 1024, 512, _lib.EPI_BIAS_GELU), "qkv": (M, 1536, 512, _lib.EPI_BIAS), "ca_q": (M, 512, 512, _lib.EPI_BIAS),
                "sty_out": (M, 512, 512, _lib.EPI_BIAS_RES)}[name]
dev = "cuda"
X = torch.randn(I, R, device=dev).to(torch.bfloat16); W = (torch.randn(J, R, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(J, device=dev); out = torch.empty(I, J, device=dev, dtype=torch.bfloat16); res = torch.randn(I, J, device=dev).to(torch.bfloat16)
d = _lib.Gemm16Desc()
d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc, d.c_f32 = X.data_ptr(), R, W.data_ptr(), R, out.data_ptr(), J, 0
d.I, d.J, d.R, d.epi, d.bias = I, J, R, epi, b.data_ptr()
if epi == _lib.EPI_BIAS_RES: d.res, d.ldr, d.res_f32 = res.data_ptr(), J, 0
lib = _lib.lib()
NB = 256
stamps = torch.zeros(8192, dtype=torch.int64, device=dev)
lib.hig_gemm_wsp16_debug_stamps(C.c_void_p(stamps.data_ptr()))
junk = torch.ones(256 << 20, device=dev)
for it in range(4):
    if not hot: junk.sum().item()
    stamps.zero_()
    _lib.check(lib.hig_gemm_bf16(C.byref(d), _lib.stream_ptr())); torch.cuda.synchronize()
    sv = stamps[4096:4096 + NB * 16].view(NB, 16).cpu()
    s = stamps[:NB * 16].view(NB, 16).cpu()
    sv = sv[s[:, 0] > 0]
    s = s[s[:, 0] > 0]
    print("%s B=%d %s run %d: %d workgroups" % (name, B, "hot" if hot else "cold", it, len(s)))
    def show(label, a, b_):
        ok = (s[:, a] > 0) & (s[:, b_] > 0)
        dlt = (s[ok, b_] - s[ok, a]).double()
        if len(dlt): print("   %-44s median %7.0f cycles   p10 %7.0f   p90 %7.0f   (n=%d)" % (label, dlt.median(), dlt.quantile(0.1), dlt.quantile(0.9), len(dlt)))
    show("start -> weights in registers", 0, 1)
    show("weights in registers -> barrier 0 (X(0))", 1, 2)
    for t in range(8):
        show("iteration %d (barrier to barrier)" % t, 2 + t, 3 + t)
    show("whole workgroup (matrix waves)", 0, 12)
    def shows(label, a, b_):
        ok = (sv[:, a] > 0) & (sv[:, b_] > 0)
        dlt = (sv[ok, b_] - sv[ok, a]).double()
        if len(dlt): print("   %-44s median %7.0f cycles   p10 %7.0f   p90 %7.0f   (n=%d)" % (label, dlt.median(), dlt.quantile(0.1), dlt.quantile(0.9), len(dlt)))
    shows("service wave, iteration 4: X DMA issue", 0, 1)
    shows("service wave, iteration 4: epilogue", 1, 2)
    shows("   res/stat DMA + pass 0 hand-off read", 1, 4)
    shows("   pass 0 arithmetic + store issue", 4, 5)
    shows("   pass 1 hand-off read", 5, 6)
    shows("   pass 1 arithmetic + store issue", 6, 7)
    shows("service wave, iteration 4: vmcnt wait", 2, 3)
lib.hig_gemm_wsp16_debug_stamps(None)
