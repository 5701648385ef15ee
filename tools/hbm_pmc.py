#!/usr/bin/env python3
"""Launches each HBM-bound kernel of the path (config-2 shapes) a few times from cold caches (1 GiB read sweep in
between) so that rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes can price its real HBM traffic.  Kernels: linear
attention ctx / apply / apply_bwd / ctx_bwd, ln_mod_silu, layernorm, ln_bwd, clip+Adam (fp32) and the bf16-storage
row / attention kernels."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from hig_amd import _lib  # noqa: E402

dev = "cuda"
B, T, H, d = 64, 196, 8, 512
hd, M = d // H, B * T
L, s = _lib.lib(), _lib.stream_ptr()
P = lambda t: t.data_ptr()
qkv = torch.randn(M, 3 * d, device=dev)
y, a_ = torch.empty(M, d, device=dev), torch.empty(M, d, device=dev)
dy = torch.randn(M, d, device=dev)
A = torch.randn(B, H, hd, hd, device=dev) * 0.1
kst, st = torch.zeros(B, d, 2, device=dev), torch.empty(M, 2, device=dev)
scr = torch.zeros(L.hig_linattn_ctx_scratch_floats(B, T, H, hd), device=dev)
bscr = torch.zeros(L.hig_linattn_bwd_scratch_floats(B, T, H, hd), device=dev)
dqkv, dA = torch.empty_like(qkv), torch.empty_like(A)
lg = torch.full((B,), T, dtype=torch.int64, device=dev)
g, be = torch.ones(d, device=dev), torch.zeros(d, device=dev)
ss = torch.randn(B, 2 * d, device=dev) * 0.1
dss = torch.zeros(B, 2 * d, device=dev)
dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
lnpart = torch.zeros(L.hig_ln_bwd_partial_floats(M, d, T), device=dev)
n_adam = 81_000_000
p_, g_, m_, v_ = (torch.randn(n_adam, device=dev) * 0.01 for _ in range(4))
v_.abs_()
nscr, gn, stp = torch.zeros(_lib.NORM_BLOCKS, device=dev), torch.zeros(1, device=dev), torch.zeros(1, device=dev, dtype=torch.int32)
q16 = qkv.to(torch.bfloat16)
y16, a16 = torch.empty(M, d, device=dev, dtype=torch.bfloat16), torch.empty(M, d, device=dev, dtype=torch.bfloat16)

At16 = torch.empty(B, H, hd, hd, device=dev, dtype=torch.bfloat16)

cases = [
    lambda: L.hig_linattn_ctx(P(qkv) + 4 * d, P(qkv) + 8 * d, 3 * d, B, T, H, hd, P(lg), P(A), P(kst), P(scr), None, s),
    lambda: L.hig_linattn_apply(P(qkv), 3 * d, P(A), P(y), d, B, T, H, hd, s),
    lambda: L.hig_linattn_apply_bwd(P(dy), d, P(qkv), 3 * d, P(A), P(dqkv), 3 * d, P(dA), B, T, H, hd, P(bscr), s),
    lambda: L.hig_linattn_ctx_bwd(P(dA), P(A), P(qkv) + 4 * d, P(qkv) + 8 * d, 3 * d, P(kst), P(lg), P(dqkv) + 4 * d, P(dqkv) + 8 * d, 3 * d, B, T, H, hd, P(bscr), s),
    lambda: L.hig_ln_mod_silu(P(y), d, M, d, P(g), P(be), P(ss), 2 * d, d, T, P(a_), d, P(st), s),
    lambda: L.hig_layernorm(P(y), d, M, d, P(g), P(be), P(a_), d, P(st), s),
    lambda: L.hig_ln_bwd(P(dy), d, P(y), d, P(st), P(g), P(be), P(ss), 2 * d, d, 1, None, 0, P(a_), d, M, d, T, P(dg), P(db), P(dss), 2 * d, P(lnpart), s),
    lambda: (L.hig_sumsq_partial(P(g_), n_adam, 1.0, P(nscr), s) or
             L.hig_clip_adam(P(p_), P(g_), P(m_), P(v_), n_adam, 2e-4, 0.9, 0.999, 1e-8, 0.5, 1.0, P(nscr), P(gn), P(stp), s)),
    lambda: L.hig_linattn_ctx_bf16(P(q16) + 2 * d, P(q16) + 4 * d, 3 * d, B, T, H, hd, P(lg), P(A), P(kst), P(scr), None, s),
    lambda: L.hig_linattn_apply_bf16(P(q16), 3 * d, P(A), P(y16), d, B, T, H, hd, s),
    lambda: L.hig_ln_bf16(P(y16), 0, d, M, d, P(g), P(be), P(ss), 2 * d, d, T, P(a16), d, s),
    lambda: L.hig_ln_bf16(P(y16), 0, d, M, d, P(g), P(be), None, 0, 0, 0, P(a16), d, s),
    # round 3: the bf16-matrix-core context build and the fused apply + stylization front (linattn16.hip)
    lambda: L.hig_linattn_ctx_mm16(P(q16) + 2 * d, P(q16) + 4 * d, 3 * d, B, T, H, hd, P(lg), P(A), P(kst), P(At16), s),
    lambda: L.hig_linattn_apply_sty_mm16(P(q16), 3 * d, P(At16), P(g), P(be), P(ss), 2 * d, d, P(a16), d, B, T, H, hd, s),
]
L.hig_layernorm(P(y), d, M, d, P(g), P(be), P(a_), d, P(st), s)     # valid stats for ln_bwd
junk = torch.ones(256 << 20, device=dev)
torch.cuda.synchronize()
for fn in cases:
    for _ in range(3):
        junk.sum().item()
        _lib.check(fn())
torch.cuda.synchronize()
print("done")
