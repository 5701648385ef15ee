"""Times the linear-attention kernels at config 2 / config 5 shapes through the C ABI (tuning aid)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hig_amd import _lib
L = _lib.lib(); s = _lib.stream_ptr(); P = lambda t: t.data_ptr()
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (B, T, H, hd) in ((64, 196, 8, 64), (32, 300, 8, 128), (64, 91, 8, 64)):
    d = H * hd
    qkv = torch.randn(B * T, 3 * d, device="cuda"); y = torch.empty(B * T, d, device="cuda"); dy = torch.randn(B * T, d, device="cuda")
    A = torch.randn(B, H, hd, hd, device="cuda") * 0.1; kst = torch.zeros(B, d, 2, device="cuda")
    scr = torch.zeros(L.hig_linattn_ctx_scratch_floats(B, T, H, hd), device="cuda")
    bscr = torch.zeros(L.hig_linattn_bwd_scratch_floats(B, T, H, hd), device="cuda")
    dqkv = torch.empty_like(qkv); dA = torch.empty_like(A)
    lg = torch.full((B,), T, dtype=torch.int64, device="cuda")
    ctx = lambda: _lib.check(L.hig_linattn_ctx(P(qkv) + 4 * d, P(qkv) + 8 * d, 3 * d, B, T, H, hd, P(lg), P(A), P(kst), P(scr), s))
    ctx()
    app = lambda: _lib.check(L.hig_linattn_apply(P(qkv), 3 * d, P(A), P(y), d, B, T, H, hd, s))
    abw = lambda: _lib.check(L.hig_linattn_apply_bwd(P(dy), d, P(qkv), 3 * d, P(A), P(dqkv), 3 * d, P(dA), B, T, H, hd, P(bscr), s))
    cbw = lambda: _lib.check(L.hig_linattn_ctx_bwd(P(dA), P(A), P(qkv) + 4 * d, P(qkv) + 8 * d, 3 * d, P(kst), P(lg), P(dqkv) + 4 * d, P(dqkv) + 8 * d, 3 * d, B, T, H, hd, P(bscr), s))
    mb = B * T * d * 4 / 1e6
    print("B=%d T=%d hd=%d (stream %.1f MB): ctx %.1f us  apply %.1f us  apply_bwd %.1f us  ctx_bwd %.1f us" % (B, T, hd, mb, t(ctx), t(app), t(abw), t(cbw)))
