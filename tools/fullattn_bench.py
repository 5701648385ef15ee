"""Time the no_eff attention kernels (hig_fullattn_fwd / _bwd) at the denoiser's shapes.
    python tools/fullattn_bench.py [B]        (HIG_FULLATTN_VALU=1 keeps head dim 64 on the VALU kernels)"""
import sys
import torch
sys.path.insert(0, ".")
import hig_amd  # noqa: F401
from hig_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T5 = int(sys.argv[2]) if len(sys.argv) > 2 else 196      # sequence length of the head-dim-128 rows (config 5: 300)
L, s, P = _lib.lib(), _lib.stream_ptr(), _lib.ptr
dev = "cuda"
for (name, Tq, Tk, H, hd, self_) in (("self d=512", 196, 196, 8, 64, True), ("cross d=512", 196, 77, 8, 64, False),
                                     ("self d=1024", T5, T5, 8, 128, True), ("cross d=1024", T5, 77, 8, 128, False)):
    d = H * hd
    q = torch.randn(B * Tq, d, device=dev)
    kv = torch.randn(B * Tk, 2 * d, device=dev)
    dy = torch.randn(B * Tq, d, device=dev)
    y, dq, dkv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(kv)
    lse, delta = torch.empty(B * H * Tq, device=dev), torch.empty(B * H * Tq, device=dev)
    ln = torch.full((B,), Tq, dtype=torch.int64, device=dev) if self_ else None

    def fwd():
        _lib.check(L.hig_fullattn_fwd(P(q), d, P(kv), kv.data_ptr() + 4 * d, 2 * d, B, Tq, Tk, H, hd, P(ln), P(y), d, P(lse), s))

    def bwd():
        _lib.check(L.hig_fullattn_bwd(P(dy), d, P(y), d, P(q), d, P(kv), kv.data_ptr() + 4 * d, 2 * d, B, Tq, Tk, H, hd,
                                      P(ln), P(lse), P(delta), P(dq), d, P(dkv), dkv.data_ptr() + 4 * d, 2 * d, s))

    q16, kv16, y16 = q.to(torch.bfloat16), kv.to(torch.bfloat16), torch.empty(B * Tq, d, device=dev, dtype=torch.bfloat16)

    def fwd16():   # bf16 storage (hig_denoiser_fwd_bf16 with no_eff): same kernel template, bf16 loads / stores
        _lib.check(L.hig_fullattn_fwd_bf16(P(q16), d, P(kv16), kv16.data_ptr() + 2 * d, 2 * d, B, Tq, Tk, H, hd, P(ln), P(y16), d, s))

    for fn, nm, mult in ((fwd, "fwd", 4), (fwd16, "fwd bf16 I/O", 4), (bwd, "bwd", 14)):   # bwd: 7 products (S and dP twice, dQ, dK, dV)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = mult * B * H * Tq * Tk * hd
        print(f"{name:14s} T={Tq} {nm}: {us:8.1f} us   {fl / us * 1e-6:7.1f} TFLOP/s executed-products", flush=True)
