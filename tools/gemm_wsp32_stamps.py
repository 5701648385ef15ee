"""Where a workgroup of the exact-fp32 weight-stationary GEMM (csrc/gemm_wsp32.hip) spends its time: s_memtime stamps of thread 0
(matrix wave 0) and thread 256 (service wave 4), and the clock the chip held (stamp span against the HIP-event time).
usage: python tools/gemm_wsp32_stamps.py [none|ffn1] [B] [hot]   (the diagnostic instances: no epilogue operand, bias + GELU)
`hot`: the same operands every run, no cache flush in between (X served by the L2 / Infinity Cache instead of HBM)."""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hig_amd import _lib
if os.environ.get("HIG_LIB_ALT"): _lib.LIB_PATH = os.environ["HIG_LIB_ALT"]   # (A/B of a variant build)
name = sys.argv[1] if len(sys.argv) > 1 else "ffn1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
hot = len(sys.argv) > 3 and sys.argv[3] == "hot"
M = B * 196
I, J, R, epi = {"ffn1": (M, 1024, 512, _lib.EPI_BIAS_GELU), "qkv": (M, 1536, 512, _lib.EPI_BIAS), "ca_q": (M, 512, 512, _lib.EPI_BIAS),
                "sty_out": (M, 512, 512, _lib.EPI_BIAS_RES), "ffn2": (M, 512, 1024, _lib.EPI_BIAS), "none": (M, 512, 512, _lib.EPI_NONE)}[name]
dev = "cuda"
X = torch.randn(I, R, device=dev); W = torch.randn(J, R, device=dev) * 0.05
b = torch.randn(J, device=dev); out = torch.empty(I, J, device=dev); res = torch.randn(I, J, device=dev)
d = _lib.GemmDesc()
d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc = X.data_ptr(), R, W.data_ptr(), R, out.data_ptr(), J
d.I, d.J, d.R, d.epi, d.bias = I, J, R, epi, b.data_ptr()
if epi == _lib.EPI_BIAS_RES: d.res, d.ldr = res.data_ptr(), J
lib = _lib.lib()
NB = 256
stamps = torch.zeros(8192, dtype=torch.int64, device=dev)
lib.hig_gemm_wsp32_debug_stamps(C.c_void_p(stamps.data_ptr()))
junk = torch.ones(256 << 20, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(4):
    if not hot: junk.sum().item()
    stamps.zero_()
    torch.cuda.synchronize()
    e0.record()
    _lib.check(lib.hig_gemm(C.byref(d), _lib.stream_ptr()))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    sv = stamps[4096:4096 + NB * 16].view(NB, 16).cpu()
    s = stamps[:NB * 16].view(NB, 16).cpu()
    sv = sv[s[:, 0] > 0]
    s = s[s[:, 0] > 0]
    span = (s[:, 12].max() - s[:, 0].min()).item() if len(s) else 0
    print("%s B=%d %s run %d: %d workgroups, %.1f us by HIP events, first start -> last end %d cycles (%.2f GHz if the two agree)" %
          (name, B, "hot" if hot else "cold", it, len(s), us, span, span / us / 1e3 if us else 0))
    def show(label, a, b_, t=s):
        ok = (t[:, a] > 0) & (t[:, b_] > 0)
        dlt = (t[ok, b_] - t[ok, a]).double()
        if len(dlt): print("   %-44s median %7.0f cycles   p10 %7.0f   p90 %7.0f   (n=%d)" % (label, dlt.median(), dlt.quantile(0.1), dlt.quantile(0.9), len(dlt)))
    if len(s):
        st0 = (s[:, 0] - s[:, 0].min()).double()
        print("   %-44s median %7.0f cycles   p10 %7.0f   p90 %7.0f" % ("workgroup start after the first one", st0.median(), st0.quantile(0.1), st0.quantile(0.9)))
    show("start -> weights in registers", 0, 1)
    show("weights in registers -> barrier 0 (X(0))", 1, 2)
    for t in range(8):
        show("iteration %d (barrier to barrier; 4096 of MFMA)" % t, 2 + t, 3 + t)
    show("matrix wave 0: wait at barrier 5", 11, 7)
    show("whole workgroup (matrix waves)", 0, 12)
    show("service wave, iteration 4: res / stats DMA", 0, 4, sv)
    show("service wave, iteration 4: X DMA issue", 4, 1, sv)
    show("service wave, iteration 4: epilogue", 1, 2, sv)
    show("service wave, iteration 4: vmcnt wait", 2, 3, sv)
lib.hig_gemm_wsp32_debug_stamps(None)
