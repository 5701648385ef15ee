"""Where one tile of the bf16 GEMM spends its time: s_memtime stamps (100 MHz ticks) of each workgroup's first tile.
usage: HIG_BF16_DBG=16 python tools/gemm16_stamps.py [shape] [B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hig_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "ffn1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
M = B * 196
I, J, R, epi = {"ffn1": (M, 1024, 512, _lib.EPI_BIAS_GELU), "qkv": (M, 1536, 512, _lib.EPI_BIAS),
                "sty_out": (M, 512, 512, _lib.EPI_BIAS_RES), "ffn2": (M, 512, 1024, _lib.EPI_BIAS)}[name]
dev = "cuda"
X = torch.randn(I, R, device=dev).to(torch.bfloat16); W = (torch.randn(J, R, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(J, device=dev); out = torch.empty(I, J, device=dev, dtype=torch.bfloat16); res = torch.randn(I, J, device=dev).to(torch.bfloat16)
d = _lib.Gemm16Desc()
d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc, d.c_f32 = X.data_ptr(), R, W.data_ptr(), R, out.data_ptr(), J, 0
d.I, d.J, d.R, d.epi, d.bias = I, J, R, epi, b.data_ptr()
if epi == _lib.EPI_BIAS_RES: d.res, d.ldr, d.res_f32 = res.data_ptr(), J, 0
lib = _lib.lib()
stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
lib.hig_gemm_bf16_debug_stamps(C.c_void_p(stamps.data_ptr()))
junk = torch.ones(256 << 20, device=dev)
for it in range(3):
    junk.sum().item(); stamps.zero_()
    _lib.check(lib.hig_gemm_bf16(C.byref(d), _lib.stream_ptr())); torch.cuda.synchronize()
    s = stamps.view(4096, 8).cpu()
    s = s[s[:, 0] > 0]
    # per-workgroup DELTAS between consecutive stamps (s_memtime ticks = shader cycles; the per-XCD counters are not
    # synchronised, so absolute offsets across workgroups mean nothing)
    cols = [0, 1, 2, 5, 3, 4, 7, 6]
    names = ["setup (coords, bias, acc = 0)", "first stage(s) of DMA issued", "main loop", "epi: pass 0 staged in LDS",
             "epi: pass 0 read, stored", "epi: pass 1 staged", "epi: pass 1 read, stored"]
    print("%s B=%d run %d: %d workgroups (first tile of each)" % (name, B, it, len(s)))
    for k in range(len(names)):
        dlt = (s[:, cols[k + 1]] - s[:, cols[k]]).double()
        print("   %-30s median %7.0f cycles   p10 %7.0f   p90 %7.0f" % (names[k], dlt.median(), dlt.quantile(0.1), dlt.quantile(0.9)))
    tot = (s[:, 6] - s[:, 0]).double()
    print("   %-30s median %7.0f cycles" % ("tile", tot.median()))
