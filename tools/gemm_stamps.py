"""Where a 64 x 64 tile of the exact-fp32 GEMM spends its time: s_memtime stamps of each workgroup's first two tiles.
usage: python tools/gemm_stamps.py [ffn1|sty_out|qkv|ffn2] [B]"""
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
X = torch.randn(I, R, device=dev); W = torch.randn(J, R, device=dev) * 0.05
b = torch.randn(J, device=dev); out = torch.empty(I, J, device=dev); res = torch.randn(I, J, device=dev)
d = _lib.GemmDesc()
d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc = X.data_ptr(), R, W.data_ptr(), R, out.data_ptr(), J
d.I, d.J, d.R, d.epi, d.bias, d.prec = I, J, R, epi, b.data_ptr(), int(os.environ.get("PREC", 0))
if epi == _lib.EPI_BIAS_RES: d.res, d.ldr = res.data_ptr(), J
lib = _lib.lib()
lib.hig_gemm_debug_stamps.argtypes = [C.c_void_p]
tail = torch.zeros(lib.hig_gemm_tail_ws_bytes(), dtype=torch.uint8, device=dev)
stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
lib.hig_gemm_debug_stamps(C.c_void_p(stamps.data_ptr()))
junk = torch.ones(256 << 20, device=dev)
for it in range(3):
    junk.sum().item(); stamps.zero_()
    _lib.check(lib.hig_gemm_ws(C.byref(d), tail.data_ptr(), tail.numel(), _lib.stream_ptr())); torch.cuda.synchronize()
s = stamps.view(4096, 8).cpu()[:, :7]
s = s[(s > 0).all(1)]
names = ["first k-tile staged", "main loop", "next tile's first fetch issued", "epi: tile staged in LDS + barrier",
         "epi: rows read, epilogue applied, stores issued", "epi: closing barrier"]
print("%s B=%d HIG_GEMM_EPI=%s: second tile of %d workgroups (cycles, median / p10 / p90)" % (name, B, os.environ.get("HIG_GEMM_EPI", "0"), len(s)))
for k in range(6):
    dl = (s[:, k + 1] - s[:, k]).double()
    print("   %-50s %7.0f %7.0f %7.0f" % (names[k], dl.median(), dl.quantile(0.1), dl.quantile(0.9)))
print("   %-50s %7.0f" % ("tile", (s[:, 6] - s[:, 0]).double().median()))
lib.hig_gemm_debug_stamps(None)
