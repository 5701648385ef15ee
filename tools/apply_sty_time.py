"""apply + stylization front: the fused kernel against the two-kernel sequence (fp32 and bf16 storage).
usage: python tools/apply_sty_time.py [B] [T] [H] [hd]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hig_amd import _lib
B, T, H, hd = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 64), (2, 196), (3, 8), (4, 64)))
d, dev = H * hd, "cuda"
L, s, P = _lib.lib(), _lib.stream_ptr(), _lib.ptr
NB = 6   # rotate operand sets (> Infinity Cache at B = 64)
qs = [torch.randn(B * T, 3 * d, device=dev) for _ in range(NB)]
A = torch.randn(B, H, hd, hd, device=dev) * 0.5
gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
ss = torch.randn(B, 2 * d, device=dev) * 0.3
outs = [torch.empty(B * T, d, device=dev) for _ in range(NB)]
ys = [torch.empty(B * T, d, device=dev) for _ in range(NB)]
st = torch.empty(B * T, 2, device=dev)
q16 = [q.to(torch.bfloat16) for q in qs]
o16 = [torch.empty(B * T, d, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
y16 = [torch.empty(B * T, d, device=dev, dtype=torch.bfloat16) for _ in range(NB)]


def fused(i):
    _lib.check(L.hig_linattn_apply_sty(P(qs[i]), 3 * d, P(A), P(gamma), P(beta), P(ss), 2 * d, d, P(outs[i]), d, B, T, H, hd, s))


def pair(i):
    _lib.check(L.hig_linattn_apply(P(qs[i]), 3 * d, P(A), P(ys[i]), d, B, T, H, hd, s))
    _lib.check(L.hig_ln_mod_silu(P(ys[i]), d, B * T, d, P(gamma), P(beta), P(ss), 2 * d, d, T, P(outs[i]), d, P(st), s))


def fused16(i):
    _lib.check(L.hig_linattn_apply_sty_bf16(P(q16[i]), 3 * d, P(A), P(gamma), P(beta), P(ss), 2 * d, d, P(o16[i]), d, B, T, H, hd, s))


def pair16(i):
    _lib.check(L.hig_linattn_apply_bf16(P(q16[i]), 3 * d, P(A), P(y16[i]), d, B, T, H, hd, s))
    _lib.check(L.hig_ln_bf16(P(y16[i]), 0, d, B * T, d, P(gamma), P(beta), P(ss), 2 * d, d, T, P(o16[i]), d, s))


for name, fn in (("fp32 fused", fused), ("fp32 apply + ln_mod_silu", pair), ("bf16 fused", fused16), ("bf16 apply + ln", pair16)):
    for i in range(NB):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(30):
        fn(i % NB)
    e1.record()
    torch.cuda.synchronize()
    print("B=%d T=%d H=%d hd=%d  %-28s %7.1f us" % (B, T, H, hd, name, e0.elapsed_time(e1) / 30 * 1e3), flush=True)
