"""Solo launch times of the bf16-storage training step's own kernels at BASELINE config 2 (HIP events, operands rotating over
several buffer sets so that nothing is served from a cache left hot by the previous launch)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hig_amd import _lib
L = _lib.lib(); dev = "cuda"; s = _lib.stream_ptr
B, T, d, H, hd, ff = 64, 196, 512, 8, 64, 1024
M = B * T; NS = 6
def bf(*shape): return [torch.randn(*shape, device=dev).to(torch.bfloat16) for _ in range(NS)]
def timeit(name, fn, bytes_, reps=30):
    for i in range(NS): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(reps): fn(i % NS)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("%-46s %8.1f us   %6.0f GB/s (algorithmic %.1f MB)" % (name, us, bytes_ / us / 1e3, bytes_ / 1e6))
da, x, res, dx = bf(M, d), bf(M, d), bf(M, d), bf(M, d)
gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
ss = 0.1 * torch.randn(B, 2 * d, device=dev)
dg, db, dss = torch.empty(d, device=dev), torch.empty(d, device=dev), torch.empty(B, 2 * d, device=dev)
part = torch.empty(L.hig_ln_bwd_partial_floats(M, d, T), device=dev)
def lnb(i, mod, with_res):
    _lib.check(L.hig_ln_bwd_bf16(_lib.ptr(da[i]), d, _lib.ptr(x[i]), 0, d, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(ss) if mod else None, 2 * d, d, int(mod),
                                 _lib.ptr(res[i]) if with_res else None, d, _lib.ptr(dx[i]), 0, d, M, d, T, _lib.ptr(dg), _lib.ptr(db),
                                 _lib.ptr(dss) if mod else None, 2 * d, _lib.ptr(part), s()))
timeit("ln_bwd16 stylization (mod, no residual)", lambda i: lnb(i, True, False), 3 * M * d * 2)
timeit("ln_bwd16 plain + residual", lambda i: lnb(i, False, True), 4 * M * d * 2)
out = bf(M, d)
def ln(i, mod):
    _lib.check(L.hig_ln_bf16(_lib.ptr(x[i]), 0, d, M, d, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(ss) if mod else None, 2 * d, d, T, _lib.ptr(out[i]), d, s()))
timeit("ln16 plain", lambda i: ln(i, False), 2 * M * d * 2)
timeit("ln16 stylization front", lambda i: ln(i, True), 2 * M * d * 2)
qkv, dqkv = bf(M, 3 * d), bf(M, 3 * d)
A = torch.randn(B, H, hd, hd, device=dev); dA = torch.empty_like(A); kst = torch.rand(B, d, 2, device=dev) + 1
bscr = torch.zeros(L.hig_linattn_bwd_scratch_floats(B, T, H, hd), device=dev)
lens = torch.full((B,), T, dtype=torch.int64, device=dev)
timeit("apply_bwd16", lambda i: _lib.check(L.hig_linattn_apply_bwd_bf16(_lib.ptr(da[i]), d, _lib.ptr(qkv[i]), 3 * d, _lib.ptr(A), _lib.ptr(dqkv[i]), 3 * d,
                                                                        _lib.ptr(dA), B, T, H, hd, _lib.ptr(bscr), s())), 3 * M * d * 2)
timeit("ctx_bwd16", lambda i: _lib.check(L.hig_linattn_ctx_bwd_bf16(_lib.ptr(dA), _lib.ptr(A), qkv[i].data_ptr() + 2 * d, qkv[i].data_ptr() + 4 * d, 3 * d,
                                                                    _lib.ptr(kst), _lib.ptr(lens), dqkv[i].data_ptr() + 2 * d, dqkv[i].data_ptr() + 4 * d, 3 * d,
                                                                    B, T, H, hd, s())), 4 * M * d * 2)
y = bf(M, d)
timeit("apply16 (fwd, fp32 MFMA, bf16 I/O)", lambda i: _lib.check(L.hig_linattn_apply_bf16(_lib.ptr(qkv[i]), 3 * d, _lib.ptr(A), _lib.ptr(y[i]), d, B, T, H, hd, s())), 2 * M * d * 2)
for J, K in ((512, 512), (1536, 512), (1024, 512), (512, 1024)):
    dC, act = bf(M, J), bf(M, K)
    dW, dbias = torch.empty(J, K, device=dev), torch.empty(J, device=dev)
    n = L.hig_wgrad_bf16_scratch_floats(J, K, 0); slabs = torch.empty(n, device=dev)
    timeit("wgrad16 J=%d K=%d (%.0f GFLOP)" % (J, K, 2 * M * J * K / 1e9),
           lambda i: _lib.check(L.hig_wgrad_bf16(_lib.ptr(dC[i]), J, _lib.ptr(act[i]), K, M, J, K, _lib.ptr(dW), _lib.ptr(dbias), 0, _lib.ptr(slabs), n, s())),
           M * (J + K) * 2)
z, f = bf(M, ff), bf(M, ff)
timeit("gelu16", lambda i: _lib.check(L.hig_gelu_bf16(_lib.ptr(z[i]), _lib.ptr(f[i]), M * ff, s())), 2 * M * ff * 2)
def gemm(i, X, Y, Cc, I, J, R, epi, resb=None):
    g = _lib.Gemm16Desc()
    g.X, g.ldx, g.Y, g.ldy, g.C, g.ldc, g.c_f32 = X[i].data_ptr(), R, Y.data_ptr(), R, Cc[i].data_ptr(), J, 0
    g.I, g.J, g.R, g.epi = I, J, R, epi
    if resb is not None: g.res, g.ldr, g.res_f32 = resb[i].data_ptr(), J, 0
    _lib.check(L.hig_gemm_bf16(C.byref(g), s()))
W512 = (torch.randn(512, 512, device=dev) * 0.05).to(torch.bfloat16)
W1t = (torch.randn(512, 1024, device=dev) * 0.05).to(torch.bfloat16)     # (d, ff): dz . W1
W2t = (torch.randn(1024, 512, device=dev) * 0.05).to(torch.bfloat16)     # (ff, d): dy3 . W2
Wqkvt = (torch.randn(512, 1536, device=dev) * 0.05).to(torch.bfloat16)
timeit("dgrad d->d plain", lambda i: gemm(i, da, W512, dx, M, 512, 512, _lib.EPI_NONE), 3 * M * d * 2 * 0 + (2 * M * d) * 2)
timeit("dgrad d->ff DGELU", lambda i: gemm(i, da, W2t, f, M, 1024, 512, _lib.EPI_DGELU, z), (M * d + 2 * M * ff) * 2)
timeit("dgrad ff->d RES", lambda i: gemm(i, z, W1t, dx, M, 512, 1024, _lib.EPI_RES, res), (M * ff + 2 * M * d) * 2)
timeit("dgrad 3d->d (tiled, K=1536)", lambda i: gemm(i, qkv, Wqkvt, dx, M, 512, 1536, _lib.EPI_NONE), (M * 3 * d + M * d) * 2)
