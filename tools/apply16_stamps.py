"""Where a workgroup of apply_sty16_kernel (csrc/linattn16.hip) spends its time: s_memtime stamps.  usage: apply16_stamps.py [B] [T H hd]"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hig_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, H, hd = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (196, 8, 64)
d = H * hd
dev = "cuda"
q16 = torch.randn(B * T, d, device=dev).to(torch.bfloat16)
At = (torch.randn(B, H, hd, hd, device=dev) * 0.5).to(torch.bfloat16)
gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
ss = 0.3 * torch.randn(B, 2 * d, device=dev)
out = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
lib = _lib.lib()
NB = ((T + 31) // 32) * B
stamps = torch.zeros(NB * 8, dtype=torch.int64, device=dev)
def run():
    _lib.check(lib.hig_linattn_apply_sty_mm16(_lib.ptr(q16), d, _lib.ptr(At), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(ss),
                                              2 * d, d, _lib.ptr(out), d, B, T, H, hd, _lib.stream_ptr()))
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): run()
e1.record(); torch.cuda.synchronize()
print("B=%d: %.2f us per launch (back to back, operands in L2)" % (B, e0.elapsed_time(e1) * 10))
lib.hig_linattn16_debug_stamps(C.c_void_p(stamps.data_ptr()))
junk = torch.ones(256 << 20, device=dev)
for it in range(3):
    if it == 0: junk.sum().item()
    stamps.zero_(); torch.cuda.synchronize()
    run(); torch.cuda.synchronize()
    s = stamps.view(NB, 8).cpu()
    names = ["requests issued + Q tile landed", "softmax", "context matrices landed (wait)", "products + row sums", "LN / SiLU -> LDS", "stores"]
    print("run %d (%s): first start -> last end %d cycles" % (it, "cold" if it == 0 else "warm", int(s[:, 6].max() - s[:, 0].min())))
    print("   start skew over workgroups: p50 %d  max %d" % (int((s[:, 0] - s[:, 0].min()).double().median()), int((s[:, 0] - s[:, 0].min()).max())))
    for k, n in enumerate(names):
        dl = (s[:, k + 1] - s[:, k]).double()
        print("   %-36s median %6.0f   p10 %6.0f   p90 %6.0f" % (n, dl.median(), dl.quantile(0.1), dl.quantile(0.9)))
    dl = (s[:, 6] - s[:, 0]).double()
    print("   %-36s median %6.0f   p10 %6.0f   p90 %6.0f" % ("whole workgroup", dl.median(), dl.quantile(0.1), dl.quantile(0.9)))
lib.hig_linattn16_debug_stamps(None)
