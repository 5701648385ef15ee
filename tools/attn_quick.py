"""apply / ctx (fp32 and bf16 I/O) at config 2, HIP events, operands rotating.  usage: attn_quick.py [B]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hig_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, H, d = 196, 8, 512
hd, M = d // H, B * T
dev = "cuda"; L, s = _lib.lib(), _lib.stream_ptr(); P = lambda t: t.data_ptr()
NS = 6
qkv = [torch.randn(M, 3 * d, device=dev) for _ in range(NS)]
y = [torch.empty(M, d, device=dev) for _ in range(NS)]
A = torch.randn(B, H, hd, hd, device=dev) * 0.1
kst = torch.zeros(B, d, 2, device=dev)
scr = torch.zeros(L.hig_linattn_ctx_scratch_floats(B, T, H, hd), device=dev)
lg = torch.full((B,), T, dtype=torch.int64, device=dev)
def timeit(fn, n=60):
    for i in range(6): fn(i % NS)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i % NS)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("B=%d  apply fp32 %.1f us   ctx fp32 %.1f us" % (B,
      timeit(lambda i: L.hig_linattn_apply(P(qkv[i]), 3 * d, P(A), P(y[i]), d, B, T, H, hd, s)),
      timeit(lambda i: L.hig_linattn_ctx(P(qkv[i]) + 4 * d, P(qkv[i]) + 8 * d, 3 * d, B, T, H, hd, P(lg), P(A), P(kst), P(scr), s))))
