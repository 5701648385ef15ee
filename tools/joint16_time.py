"""joint_embed16 alone: back to back on the same operands (L2-warm) and rotating over operand sets.  usage: joint16_time.py [B]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hig_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, F, d = 196, 150, 512
M = B * T
dev = "cuda"; L, s = _lib.lib(), _lib.stream_ptr()
NS = 8
x = [torch.randn(M, F, device=dev) for _ in range(NS)]
out = [torch.empty(M, d, device=dev, dtype=torch.bfloat16) for _ in range(NS)]
Fp = (F + 31) // 32 * 32
W = torch.zeros(d, Fp, device=dev, dtype=torch.bfloat16); W[:, :F] = (torch.randn(d, F, device=dev) * 0.1).to(torch.bfloat16)
b = torch.randn(d, device=dev); pos = torch.randn(T, d, device=dev)
def run(i):
    _lib.check(L.hig_joint_embed_bf16_w(_lib.ptr(x[i]), M, F, _lib.ptr(W), _lib.ptr(b), _lib.ptr(pos), d, T, 0, _lib.ptr(out[i]), d, d, s))
def timeit(sel, n=80):
    for k in range(8): run(sel(k))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n): run(sel(k))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("B=%d: same operands %.1f us   rotating over %d sets %.1f us" % (B, timeit(lambda k: 0), NS, timeit(lambda k: k % NS)))
