"""Per-kernel parity: every C-ABI entry point of libhig.so against a plain PyTorch fp32/fp64 CPU
statement of the same op (for floating point the tolerance is written at each assert).
All tests need the MI355X: run through `gpurun -- python -m pytest tests -m gpu`."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import hig_amd  # noqa: E402
from hig_amd import _lib  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale)


_KEEP = []  # device tensors created inline must outlive the launch that reads them


def P(t):
    if t is None:
        return None
    _KEEP.append(t)
    if len(_KEEP) > 256:
        del _KEEP[:128]
    return t.data_ptr()


def gemm(X, Y, I, J, R, x_rs=0, y_rs=0, xf=0, xf_on_y=0, epi=0, bias=None, res=None, aux=None,
         stats=None, gamma=None, beta=None, ss=None, ss_shift_off=0, rows_per_sample=0, pos=None, T=0,
         prec=0, tail_ws=None):
    out = torch.full((I, J), float("nan"), device=DEV)
    d = _lib.GemmDesc()
    d.X, d.ldx, d.x_rs = P(X), X.stride(0), x_rs
    d.Y, d.ldy, d.y_rs = P(Y), Y.stride(0), y_rs
    d.C, d.ldc = P(out), J
    d.I, d.J, d.R = I, J, R
    d.xf, d.xf_on_y, d.epi, d.prec = xf, xf_on_y, epi, prec
    d.bias = P(bias)
    if res is not None:
        d.res, d.ldr = P(res), res.stride(0)
    if aux is not None:
        d.aux, d.ldaux = P(aux), aux.stride(0)
    d.stats, d.gamma, d.beta = P(stats), P(gamma), P(beta)
    if ss is not None:
        d.ss, d.ss_ld, d.ss_shift_off, d.rows_per_sample = P(ss), ss.stride(0), ss_shift_off, rows_per_sample
    if pos is not None:
        d.pos, d.ldpos, d.T = P(pos), pos.stride(0), T
    if tail_ws is not None:
        _lib.check(_lib.lib().hig_gemm_ws(C.byref(d), P(tail_ws), tail_ws.numel(), _lib.stream_ptr()))
    else:
        _lib.check(_lib.lib().hig_gemm(C.byref(d), _lib.stream_ptr()))
    torch.cuda.synchronize()
    return out


def stats_of(x):
    m = x.mean(-1)
    v = x.var(-1, unbiased=False)
    return torch.stack([m, torch.rsqrt(v + 1e-5)], -1).contiguous()


@pytest.mark.parametrize("I,J,R", [(128, 128, 32), (130, 150, 150), (64, 64, 512), (300, 1536, 512), (2, 2048, 64)])
def test_gemm_forward_bias(I, J, R):
    X, Y, b = rnd(I, R), rnd(J, R, seed=1), rnd(J, seed=2)
    out = gemm(X.to(DEV), Y.to(DEV), I, J, R, epi=_lib.EPI_BIAS, bias=b.to(DEV))
    ref = X.double() @ Y.double().T + b.double()
    assert rel(out, ref) < 2e-6  # fp32 MFMA, fp32 accumulate


@pytest.mark.parametrize("I,J,R", [(256, 256, 512), (12544 // 8, 1024, 512), (300, 1536, 512), (70, 64, 2048)])
def test_gemm_split_bf16_modes(I, J, R):
    """HIG_PREC_BF16X3: hi/lo split of both operands, 3 bf16 MFMAs, fp32 accumulate -> ~2^-16
    per product; HIG_PREC_BF16: single product (bf16 rounding of the operands)."""
    X, Y, b = rnd(I, R), rnd(J, R, seed=1), rnd(J, seed=2)
    ref = X.double() @ Y.double().T + b.double()
    out3 = gemm(X.to(DEV), Y.to(DEV), I, J, R, epi=_lib.EPI_BIAS, bias=b.to(DEV), prec=_lib.PREC_BF16X3)
    assert rel(out3, ref) < 2e-5
    out1 = gemm(X.to(DEV), Y.to(DEV), I, J, R, epi=_lib.EPI_BIAS, bias=b.to(DEV), prec=_lib.PREC_BF16)
    assert 1e-4 < rel(out1, ref) < 1e-2
    # bf16-representable small integers are exact in both modes
    g = torch.Generator().manual_seed(5)
    Xi = torch.randint(-8, 9, (I, R), generator=g).float()
    Yi = torch.randint(-8, 9, (J, R), generator=g).float()
    for prec in (_lib.PREC_BF16X3, _lib.PREC_BF16):
        assert torch.equal(gemm(Xi.to(DEV), Yi.to(DEV), I, J, R, prec=prec).cpu(), Xi @ Yi.T)
    # fused prologue + epilogue still apply
    gm, be = 1 + 0.1 * rnd(R, seed=3), 0.1 * rnd(R, seed=4)
    out = gemm(X.to(DEV), Y.to(DEV), I, J, R, xf=_lib.XF_LN, epi=_lib.EPI_BIAS, bias=b.to(DEV),
               stats=stats_of(X).to(DEV), gamma=gm.to(DEV), beta=be.to(DEV), prec=_lib.PREC_BF16X3)
    refln = F.linear(F.layer_norm(X.double(), (R,), gm.double(), be.double()), Y.double(), b.double())
    assert rel(out, refln) < 2e-5


@pytest.mark.parametrize("I,J,R,epi", [(12544, 512, 512, "res"), (12544, 1024, 512, "gelu"), (12544, 1536, 512, "bias"),
                                       (12544, 512, 1024, "bias"), (8000, 320, 256, "res"), (6272, 512, 512, "bias"),
                                       (12544, 512, 96, "bias"), (300, 512, 512, "bias")])
def test_gemm_split_tail(I, J, R, epi):
    """hig_gemm_ws: the tiles of the last, partly filled round are cut along the reduce range and finished by the
    workgroup that draws the last ticket.  Same numbers as the fp64 product, run-to-run identical, integer operands
    exact, tickets left zero, rows outside the tail tiles bit-identical to hig_gemm."""
    L = _lib.lib()
    ws = torch.full((L.hig_gemm_tail_ws_bytes(),), 0x5A, dtype=torch.uint8, device=DEV)
    ws[:1024] = 0
    X, Y, b, r = rnd(I, R), rnd(J, R, seed=1), rnd(J, seed=2), rnd(I, J, seed=3)
    kw = dict(bias=b.to(DEV))
    ref = X.double() @ Y.double().T + b.double()
    if epi == "res":
        kw.update(epi=_lib.EPI_BIAS_RES, res=r.to(DEV))
        ref = ref + r.double()
    elif epi == "gelu":
        kw.update(epi=_lib.EPI_BIAS_GELU)
        ref = F.gelu(ref)
    else:
        kw.update(epi=_lib.EPI_BIAS)
    Xg, Yg = X.to(DEV), Y.to(DEV)
    plain = gemm(Xg, Yg, I, J, R, **kw)
    a = gemm(Xg, Yg, I, J, R, tail_ws=ws, **kw)
    assert rel(a, ref) < 2e-6
    assert int(ws[:1024].max()) == 0                      # every ticket counter is back to zero
    for _ in range(3):
        assert torch.equal(gemm(Xg, Yg, I, J, R, tail_ws=ws, **kw), a)
    # Stale parked sums must never be read: alternate between two DIFFERENT operand sets on the same scratch, with the
    # parked-sum area poisoned (NaN bytes) before every launch.  A ticket counted before another wave's partial sums
    # have landed would make the last arriver add poison (or the other operand set's sums) into the tile.
    X2g, Y2g = rnd(I, R, seed=11).to(DEV), rnd(J, R, seed=12).to(DEV)
    a2 = None
    for it in range(12):
        ws[1024:] = 0xFF
        if it % 2 == 0:
            assert torch.equal(gemm(Xg, Yg, I, J, R, tail_ws=ws, **kw), a)
        else:
            o2 = gemm(X2g, Y2g, I, J, R, tail_ws=ws, **kw)
            assert torch.isfinite(o2).all()
            a2 = o2 if a2 is None else a2
            assert torch.equal(o2, a2)
    assert int(ws[:1024].max()) == 0
    tiles = ((I + 63) // 64) * ((J + 63) // 64)
    main_rows = (tiles - tiles % 256) // ((J + 63) // 64) * 64 if tiles > 256 else I
    assert torch.equal(a[:main_rows], plain[:main_rows])   # whole rounds: untouched by the split
    g = torch.Generator().manual_seed(5)
    Xi = torch.randint(-4, 5, (I, R), generator=g).float()
    Yi = torch.randint(-4, 5, (J, R), generator=g).float()
    assert torch.equal(gemm(Xi.to(DEV), Yi.to(DEV), I, J, R, tail_ws=ws).cpu(), Xi @ Yi.T)


def test_gemm_exact_small_integers():
    # integer-valued operands: the fp32 MFMA path must be EXACT (catches layout / k-order bugs)
    I, J, R = 192, 160, 96
    g = torch.Generator().manual_seed(3)
    X = torch.randint(-4, 5, (I, R), generator=g).float()
    Y = torch.randint(-4, 5, (J, R), generator=g).float()  # asymmetric
    out = gemm(X.to(DEV), Y.to(DEV), I, J, R)
    assert torch.equal(out.cpu(), X @ Y.T)


def test_gemm_ln_prologue_and_gelu_epilogue():
    I, J, R = 200, 96, 128
    X, Y, b = rnd(I, R, scale=2.0) + 0.5, rnd(J, R, seed=1, scale=0.1), rnd(J, seed=2)
    g, be = 1 + 0.1 * rnd(R, seed=3), 0.1 * rnd(R, seed=4)
    out = gemm(X.to(DEV), Y.to(DEV), I, J, R, xf=_lib.XF_LN, epi=_lib.EPI_BIAS, bias=b.to(DEV),
               stats=stats_of(X).to(DEV), gamma=g.to(DEV), beta=be.to(DEV))
    ref = F.linear(F.layer_norm(X.double(), (R,), g.double(), be.double()), Y.double(), b.double())
    assert rel(out, ref) < 5e-6
    aux = torch.zeros(I, J, device=DEV)
    out = gemm(X.to(DEV), Y.to(DEV), I, J, R, epi=_lib.EPI_BIAS_GELU, bias=b.to(DEV), aux=aux)
    z = F.linear(X.double(), Y.double(), b.double())
    assert rel(aux, z) < 2e-6 and rel(out, F.gelu(z)) < 5e-6


def test_gemm_stylization_prologue_residual():
    B, T, d, J = 3, 50, 64, 64
    I, R = B * T, d
    X, Y, b = rnd(I, R), rnd(J, R, seed=1, scale=0.2), rnd(J, seed=2)
    g, be = 1 + 0.1 * rnd(R, seed=3), 0.1 * rnd(R, seed=4)
    ss = rnd(B, 6 * d, seed=5, scale=0.5)   # block 1 of 3: scale at [2d,3d), shift at [3d,4d)
    res = rnd(I, J, seed=6)
    out = gemm(X.to(DEV), Y.to(DEV), I, J, R, xf=_lib.XF_LN_MOD_SILU, epi=_lib.EPI_BIAS_RES,
               bias=b.to(DEV), res=res.to(DEV), stats=stats_of(X).to(DEV), gamma=g.to(DEV),
               beta=be.to(DEV), ss=ss.to(DEV)[:, 2 * d:], ss_shift_off=d, rows_per_sample=T)
    n = F.layer_norm(X.double(), (R,), g.double(), be.double()).view(B, T, d)
    u = n * (1 + ss.double()[:, None, 2 * d:3 * d]) + ss.double()[:, None, 3 * d:4 * d]
    ref = res.double() + F.linear(F.silu(u).view(I, d), Y.double(), b.double())
    assert rel(out, ref) < 5e-6


def test_gemm_silu_prologue_and_pos_epilogue():
    I, J, R, T = 96, 80, 64, 32
    X, Y, b = rnd(I, R), rnd(J, R, seed=1), rnd(J, seed=2)
    out = gemm(X.to(DEV), Y.to(DEV), I, J, R, xf=_lib.XF_SILU, epi=_lib.EPI_BIAS, bias=b.to(DEV))
    assert rel(out, F.linear(F.silu(X.double()), Y.double(), b.double())) < 5e-6
    pos = rnd(40, J, seed=7)
    out = gemm(X.to(DEV), Y.to(DEV), I, J, R, epi=_lib.EPI_BIAS_POS, bias=b.to(DEV), pos=pos.to(DEV), T=T)
    ref = F.linear(X.double(), Y.double(), b.double()) + pos.double()[:T].repeat(I // T, 1)
    assert rel(out, ref) < 2e-6


@pytest.mark.parametrize("I,J,R", [(200, 150, 64), (128, 128, 1536), (70, 512, 150)])
def test_gemm_dgrad(I, J, R):
    dC, W = rnd(I, R), rnd(R, J, seed=1)          # W is (out=R, in=J): dA = dC @ W
    out = gemm(dC.to(DEV), W.to(DEV), I, J, R, y_rs=1)
    assert rel(out, dC.double() @ W.double()) < 2e-6
    res = rnd(I, J, seed=2)
    out = gemm(dC.to(DEV), W.to(DEV), I, J, R, y_rs=1, epi=_lib.EPI_RES, res=res.to(DEV))
    assert rel(out, res.double() + dC.double() @ W.double()) < 2e-6
    z = rnd(I, J, seed=3)
    zd = z.double().requires_grad_(True)
    F.gelu(zd).backward(dC.double() @ W.double())
    out = gemm(dC.to(DEV), W.to(DEV), I, J, R, y_rs=1, epi=_lib.EPI_DGELU, aux=z.to(DEV))
    assert rel(out, zd.grad) < 5e-6


def test_gemm_wgrad_with_fused_activation_recompute():
    B, T, d, n_out = 2, 70, 64, 96
    M = B * T
    dC, y = rnd(M, n_out), rnd(M, d, seed=1)
    out = gemm(dC.to(DEV), y.to(DEV), n_out, d, M, x_rs=1, y_rs=1)
    assert rel(out, dC.double().T @ y.double()) < 2e-6
    g, be = 1 + 0.1 * rnd(d, seed=3), 0.1 * rnd(d, seed=4)
    ss = rnd(B, 2 * d, seed=5, scale=0.5)
    n = F.layer_norm(y.double(), (d,), g.double(), be.double())
    out = gemm(dC.to(DEV), y.to(DEV), n_out, d, M, x_rs=1, y_rs=1, xf=_lib.XF_LN, xf_on_y=1,
               stats=stats_of(y).to(DEV), gamma=g.to(DEV), beta=be.to(DEV))
    assert rel(out, dC.double().T @ n) < 5e-6
    a = F.silu(n.view(B, T, d) * (1 + ss.double()[:, None, :d]) + ss.double()[:, None, d:]).view(M, d)
    out = gemm(dC.to(DEV), y.to(DEV), n_out, d, M, x_rs=1, y_rs=1, xf=_lib.XF_LN_MOD_SILU, xf_on_y=1,
               stats=stats_of(y).to(DEV), gamma=g.to(DEV), beta=be.to(DEV), ss=ss.to(DEV),
               ss_shift_off=d, rows_per_sample=T)
    assert rel(out, dC.double().T @ a) < 5e-6
    out = gemm(dC.to(DEV), y.to(DEV), n_out, d, M, x_rs=1, y_rs=1, xf=_lib.XF_SILU, xf_on_y=1)
    assert rel(out, dC.double().T @ F.silu(y.double())) < 5e-6


@pytest.mark.parametrize("rows,n", [(7, 64), (1000, 512), (33, 1024), (5, 150), (77 * 2, 256)])
def test_rowstats(rows, n):
    x = rnd(rows, n, scale=3.0) + 1.5
    st = torch.zeros(rows, 2, device=DEV)
    _lib.check(_lib.lib().hig_rowstats(P(x.to(DEV)), n, rows, n, P(st), _lib.stream_ptr()))
    assert rel(st, stats_of(x.double())) < 1e-6


@pytest.mark.parametrize("B,T,n", [(2, 16, 64), (3, 50, 512), (2, 9, 1024), (5, 7, 256)])
def test_ln_mod_silu_rows(B, T, n):
    M = B * T
    x = rnd(M, n, scale=2.0) + 0.3
    g, be, ss = 1 + 0.1 * rnd(n, seed=2), 0.1 * rnd(n, seed=3), rnd(B, 6 * n, seed=4, scale=0.5)
    a, st = torch.zeros(M, n, device=DEV), torch.zeros(M, 2, device=DEV)
    ssg = ss.to(DEV)
    _lib.check(_lib.lib().hig_ln_mod_silu(P(x.to(DEV)), n, M, n, P(g.to(DEV)), P(be.to(DEV)),
                                          ssg.data_ptr() + 4 * 2 * n, 6 * n, n, T, P(a), n, P(st), _lib.stream_ptr()))
    nrm = F.layer_norm(x.double(), (n,), g.double(), be.double()).view(B, T, n)
    ref = F.silu(nrm * (1 + ss.double()[:, None, 2 * n:3 * n]) + ss.double()[:, None, 3 * n:4 * n]).view(M, n)
    assert rel(a, ref) < 2e-6 and rel(st, stats_of(x.double())) < 1e-6


@pytest.mark.parametrize("mod", [0, 1])
@pytest.mark.parametrize("B,T,n", [(2, 16, 64), (3, 50, 512), (2, 77, 256), (2, 9, 1024)])
def test_ln_bwd(mod, B, T, n):
    M = B * T
    x = rnd(M, n, scale=2.0) + 0.3
    da, g, be = rnd(M, n, seed=1), 1 + 0.1 * rnd(n, seed=2), 0.1 * rnd(n, seed=3)
    ss = rnd(B, 2 * n, seed=4, scale=0.5)
    res = rnd(M, n, seed=5)
    xd, gd, bd, ssd = (v.double().requires_grad_(True) for v in (x, g, be, ss))
    nrm = F.layer_norm(xd, (n,), gd, bd)
    if mod:
        a = F.silu(nrm.view(B, T, n) * (1 + ssd[:, None, :n]) + ssd[:, None, n:]).view(M, n)
    else:
        a = nrm
    a.backward(da.double())
    L = _lib.lib()
    npart = L.hig_ln_bwd_partial_floats(M, n, T)
    part = torch.zeros(npart, device=DEV)
    dx, dg, db = torch.zeros(M, n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    dss = torch.zeros(B, 2 * n, device=DEV)
    xg, ssg, resg = x.to(DEV), ss.to(DEV), res.to(DEV)
    _lib.check(L.hig_ln_bwd(P(da.to(DEV)), n, P(xg), n, P(stats_of(x).to(DEV)), P(g.to(DEV)), P(be.to(DEV)),
                            P(ssg) if mod else None, 2 * n, n, mod, P(resg), n, P(dx), n, M, n, T, P(dg), P(db),
                            P(dss) if mod else None, 2 * n, P(part), _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert rel(dx, res.double() + xd.grad) < 1e-5
    assert rel(dg, gd.grad) < 1e-5 and rel(db, bd.grad) < 1e-5
    if mod:
        assert rel(dss, ssd.grad) < 1e-5


@pytest.mark.parametrize("rows,cols,ln", [(130, 70, False), (12544 // 4, 512, True), (64, 64, False), (77, 256, True)])
def test_transpose_with_optional_layernorm(rows, cols, ln):
    x = rnd(rows, cols, scale=2.0) + 0.5
    g, be = 1 + 0.1 * rnd(cols, seed=2), 0.1 * rnd(cols, seed=3)
    out = torch.full((cols, rows), float("nan"), device=DEV)
    _lib.check(_lib.lib().hig_transpose(P(x.to(DEV)), cols, rows, cols, P(out), rows,
                                        P(stats_of(x).to(DEV)) if ln else None, P(g.to(DEV)) if ln else None,
                                        P(be.to(DEV)) if ln else None, _lib.stream_ptr()))
    ref = F.layer_norm(x.double(), (cols,), g.double(), be.double()) if ln else x.double()
    if ln:
        assert rel(out, ref.T) < 2e-6
    else:
        assert torch.equal(out.cpu(), x.T)


@pytest.mark.parametrize("rows,n", [(1000, 512), (64, 24576 // 8), (77, 150), (3, 5)])
def test_colsum(rows, n):
    x = rnd(rows, n)
    out = torch.zeros(n, device=DEV)
    part = torch.zeros(_lib.COLSUM_CHUNKS * n, device=DEV)
    _lib.check(_lib.lib().hig_colsum(P(x.to(DEV)), n, rows, n, P(out), P(part), _lib.stream_ptr()))
    assert rel(out, x.double().sum(0)) < 1e-6


def _linattn_ref(Q, K, V, lengths, H):
    B, T, D = Q.shape
    mask = (torch.arange(T)[None] < lengths[:, None]).double().unsqueeze(-1)
    q = F.softmax(Q.view(B, T, H, -1), dim=-1)
    k = F.softmax((K + (1 - mask) * -1000000).view(B, T, H, -1), dim=1)
    v = (V * mask).view(B, T, H, -1)
    A = torch.einsum("bnhd,bnhl->bhdl", k, v)
    y = torch.einsum("bnhd,bhdl->bnhl", q, A).reshape(B, T, D)
    return y, A


@pytest.mark.parametrize("B,T,H,hd,lens", [(2, 16, 8, 8, (16, 9)), (2, 60, 8, 16, (60, 41)),
                                           (3, 196, 8, 64, (196, 77, 1)), (2, 300, 2, 128, (300, 123)),
                                           (2, 70, 4, 32, (70, 64)),
                                           # B * H >= 256: the chunk-walking / online-softmax kernels, ragged incl. empty
                                           (32, 130, 8, 64, tuple((17 * i) % 131 for i in range(32))),
                                           (32, 70, 8, 128, tuple((29 * i) % 71 for i in range(32)))])
def test_linear_attention_forward_backward(B, T, H, hd, lens):
    d = H * hd
    qkv = rnd(B * T, 3 * d, scale=1.5)
    dy = rnd(B * T, d, seed=1)
    lengths = torch.tensor(lens)
    qd = qkv.double().view(B, T, 3 * d).requires_grad_(True)
    y_ref, A_ref = _linattn_ref(qd[..., :d], qd[..., d:2 * d], qd[..., 2 * d:], lengths, H)
    y_ref.backward(dy.double().view(B, T, d))
    L = _lib.lib()
    g = qkv.to(DEV)
    A = torch.zeros(B, H, hd, hd, device=DEV)
    kst = torch.zeros(B, d, 2, device=DEV)
    y = torch.zeros(B * T, d, device=DEV)
    lg = lengths.to(DEV)
    s = _lib.stream_ptr()
    # with scratch: chunk-parallel build (hd 64 / 128, more than one 64-row chunk); without: one workgroup per (b, h)
    for use_scratch in (True, False):
        scr = torch.zeros(L.hig_linattn_ctx_scratch_floats(B, T, H, hd), device=DEV) if use_scratch else None
        A.zero_()
        kst.zero_()
        _lib.check(L.hig_linattn_ctx(g.data_ptr() + 4 * d, g.data_ptr() + 8 * d, 3 * d, B, T, H, hd, P(lg), P(A), P(kst),
                                     P(scr), s))
        assert rel(A, A_ref) < 1e-5, (use_scratch, rel(A, A_ref))
    _lib.check(L.hig_linattn_apply(P(g), 3 * d, P(A), P(y), d, B, T, H, hd, s))
    torch.cuda.synchronize()
    assert rel(A, A_ref) < 5e-6      # fp32 exp/sum vs fp64
    assert rel(y, y_ref.reshape(B * T, d)) < 5e-6
    dqkv = torch.full((B * T, 3 * d), float("nan"), device=DEV)
    dA = torch.zeros(B, H, hd, hd, device=DEV)
    scr = torch.zeros(L.hig_linattn_bwd_scratch_floats(B, T, H, hd), device=DEV)
    _lib.check(L.hig_linattn_apply_bwd(P(dy.to(DEV)), d, P(g), 3 * d, P(A), P(dqkv), 3 * d, P(dA), B, T, H, hd, P(scr), s))
    _lib.check(L.hig_linattn_ctx_bwd(P(dA), P(A), g.data_ptr() + 4 * d, g.data_ptr() + 8 * d, 3 * d, P(kst), P(lg),
                                     dqkv.data_ptr() + 4 * d, dqkv.data_ptr() + 8 * d, 3 * d, B, T, H, hd, P(scr), s))
    torch.cuda.synchronize()
    ref = qd.grad.view(B * T, 3 * d)
    assert rel(dqkv[:, :d], ref[:, :d]) < 2e-5
    assert rel(dqkv[:, 2 * d:], ref[:, 2 * d:]) < 2e-5
    # dK is a difference of near-equal terms (column softmax): absolute floor relative to dV scale
    err = (dqkv[:, d:2 * d].double().cpu() - ref[:, d:2 * d]).norm()
    assert err < 2e-5 * ref[:, 2 * d:].norm() + 2e-5 * ref[:, d:2 * d].norm()


@pytest.mark.parametrize("B,Tq,Tk,H,hd,lens", [(2, 16, 16, 8, 8, (16, 9)), (2, 60, 60, 8, 16, (60, 41)),
                                                (2, 196, 196, 4, 64, (196, 77)), (2, 70, 77, 4, 32, None),
                                                (1, 130, 77, 2, 64, None), (1, 300, 300, 2, 128, (211,)),
                                                (2, 130, 77, 2, 128, None), (3, 257, 129, 8, 64, (257, 1, 100))])
def test_full_attention_forward_backward(B, Tq, Tk, H, hd, lens):
    """no_eff attention (transformer.py:208-227,242-262) incl. the query-axis -1e5 mask, vs fp64."""
    d = H * hd
    q, kv, dy = rnd(B * Tq, d, scale=1.5), rnd(B * Tk, 2 * d, seed=1, scale=1.5), rnd(B * Tq, d, seed=2)
    qd = q.double().view(B, Tq, H, hd).requires_grad_(True)
    kvd = kv.double().view(B, Tk, 2 * d).requires_grad_(True)
    kd, vd = kvd[..., :d].reshape(B, Tk, H, hd), kvd[..., d:].reshape(B, Tk, H, hd)
    att = torch.einsum("bnhd,bmhd->bnmh", qd, kd) / math.sqrt(hd)
    valid = torch.ones(B, Tq, dtype=torch.bool)
    if lens is not None:
        valid = torch.arange(Tq)[None] < torch.tensor(lens)[:, None]
        # fp32 semantics of the reference: logits + (-1e5) rounded to fp32 on padded QUERY rows
        att = torch.where(valid[:, :, None, None], att, (att.float() + (-100000.0)).double() + 0 * att)
    w = torch.softmax(att, dim=2)
    y_ref = torch.einsum("bnmh,bmhd->bnhd", w, vd).reshape(B, Tq, d)
    (y_ref * dy.double().view(B, Tq, d) * valid[..., None]).sum().backward()
    L, s = _lib.lib(), _lib.stream_ptr()
    qg, kvg = q.to(DEV), kv.to(DEV)
    lg = None if lens is None else torch.tensor(lens).to(DEV)
    y = torch.full((B * Tq, d), float("nan"), device=DEV)
    lse = torch.zeros(B * H * Tq, device=DEV)
    _lib.check(L.hig_fullattn_fwd(P(qg), d, P(kvg), kvg.data_ptr() + 4 * d, 2 * d, B, Tq, Tk, H, hd, P(lg), P(y), d, P(lse), s))
    torch.cuda.synchronize()
    vm = valid.reshape(-1)
    assert rel(y[vm.to(DEV)], y_ref.reshape(B * Tq, d)[vm]) < 5e-6
    if lens is not None and (~vm).any():   # padded query rows: logits quantised to 2^-7 by the -1e5 offset
        assert rel(y[(~vm).to(DEV)], y_ref.reshape(B * Tq, d)[~vm]) < 2e-2
    dyg = (dy * vm[:, None]).to(DEV)
    dq, dkv = torch.full((B * Tq, d), float("nan"), device=DEV), torch.full((B * Tk, 2 * d), float("nan"), device=DEV)
    delta = torch.zeros(B * H * Tq, device=DEV)
    _lib.check(L.hig_fullattn_bwd(P(dyg), d, P(y), d, P(qg), d, P(kvg), kvg.data_ptr() + 4 * d, 2 * d, B, Tq, Tk, H, hd,
                                  P(lg), P(lse), P(delta), P(dq), d, P(dkv), dkv.data_ptr() + 4 * d, 2 * d, s))
    torch.cuda.synchronize()
    assert rel(dq, qd.grad.reshape(B * Tq, d)) < 2e-5
    assert rel(dkv, kvd.grad.reshape(B * Tk, 2 * d)) < 2e-5


@pytest.mark.parametrize("hd,H,B,T", [(64, 8, 3, 196), (128, 8, 2, 75), (64, 4, 2, 33), (128, 4, 1, 300)])
def test_fused_apply_stylization_front_fp32(hd, H, B, T):
    """hig_linattn_apply_sty (the inference forward's attention epilogue: apply + LayerNorm + (1 + scale) + shift + SiLU in
    one kernel, transformer.py:111,116-118 then :81-85) == hig_linattn_apply followed by hig_ln_mod_silu, and both == fp64."""
    d = H * hd
    g = torch.Generator().manual_seed(hd + H + B + T)
    q = (torch.randn(B * T, 3 * d, generator=g) * 2).to(DEV)          # queries inside a q/k/v buffer (ld = 3 d)
    A = (torch.randn(B, H, hd, hd, generator=g) * 0.5).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV), (0.1 * torch.randn(d, generator=g)).to(DEV)
    ss = (0.3 * torch.randn(B, 6 * d, generator=g)).to(DEV)            # one (scale, shift) pair inside the stacked table
    L, s = _lib.lib(), _lib.stream_ptr()
    out = torch.full((B * T, d), float("nan"), device=DEV)
    _lib.check(L.hig_linattn_apply_sty(P(q), 3 * d, P(A), P(gamma), P(beta), ss.data_ptr() + 8 * d, 6 * d, d, P(out), d,
                                       B, T, H, hd, s))
    y = torch.empty(B * T, d, device=DEV)
    two = torch.empty(B * T, d, device=DEV)
    st = torch.empty(B * T, 2, device=DEV)
    _lib.check(L.hig_linattn_apply(P(q), 3 * d, P(A), P(y), d, B, T, H, hd, s))
    _lib.check(L.hig_ln_mod_silu(P(y), d, B * T, d, P(gamma), P(beta), ss.data_ptr() + 8 * d, 6 * d, d, T, P(two), d, P(st), s))
    torch.cuda.synchronize()
    qd = q[:, :d].double().cpu().view(B, T, H, hd)
    yd = torch.einsum("bthc,bhcl->bthl", torch.softmax(qd, -1), A.double().cpu()).reshape(B * T, d)
    ref = F.layer_norm(yd, (d,), gamma.double().cpu(), beta.double().cpu(), 1e-5)
    sc = ss[:, 2 * d:3 * d].double().cpu().repeat_interleave(T, 0)
    sh = ss[:, 3 * d:4 * d].double().cpu().repeat_interleave(T, 0)
    ref = F.silu(ref * (1 + sc) + sh)
    assert torch.isfinite(out).all()
    assert rel(out, ref) < 2e-6 and rel(two, ref) < 2e-6
    assert rel(out, two) < 2e-6


def test_timestep_embedding_matches_reference_formula():
    from oracle import denoiser_ref as R
    t = torch.tensor([0, 1, 7, 500, 999])
    for d in (64, 128, 512):
        out = torch.zeros(len(t), d, device=DEV)
        _lib.check(_lib.lib().hig_timestep_embedding(P(t.to(DEV)), len(t), d, P(out), _lib.stream_ptr()))
        ref = R.timestep_embedding(t, d)
        assert (out.cpu() - ref).abs().max() < 2e-4  # |arg| <= 999: one fp32 ulp of arg is 6e-5
        # the bf16 entry of the bf16-storage forward: the same values, rounded once
        out16 = torch.zeros(len(t), d, device=DEV, dtype=torch.bfloat16)
        _lib.check(_lib.lib().hig_timestep_embedding_bf16(P(t.to(DEV)), len(t), d, P(out16), _lib.stream_ptr()))
        assert torch.equal(out16, out.to(torch.bfloat16))


def test_ddpm_elementwise_against_golden(gold):
    from oracle import diffusion_ref as D
    from oracle import fill
    g = gold("g4_diffusion.npz")
    gd = hig_amd.GaussianDiffusion(
        betas=hig_amd.models.gaussian_diffusion.get_named_beta_schedule("linear", 1000),
        model_mean_type=hig_amd.models.gaussian_diffusion.ModelMeanType.EPSILON,
        model_var_type=hig_amd.models.gaussian_diffusion.ModelVarType.FIXED_SMALL,
        loss_type=hig_amd.models.gaussian_diffusion.LossType.MSE)
    x, eps, t = (torch.tensor(g[k]).to(DEV) for k in ("x", "eps", "t"))
    z0 = (fill.tensor_for("g4.z.0", x.shape) * 10.0).to(DEV)
    z1 = (fill.tensor_for("g4.z.1", x.shape) * 10.0).to(DEV)
    xt = gd.q_sample(x, t, noise=z0)
    assert rel(xt, torch.tensor(g["q_sample"])) < 1e-6
    B = x.shape[0]
    sample, pred = torch.empty_like(x), torch.empty_like(x)
    _lib.check(_lib.lib().hig_p_sample_step(P(x), P(eps), P(z1), P(t), P(gd.device_table(x.device)), 1000, B,
                                            x.numel() // B, P(sample), P(pred), _lib.stream_ptr()))
    assert rel(pred, torch.tensor(g["pred_xstart"])) < 1e-6
    assert rel(sample, torch.tensor(g["p_sample"])) < 1e-6
    # t == 0 row: no noise
    assert rel(sample[0], torch.tensor(g["mean"])[0]) < 1e-6
    # device table equals the float64 tables cast to fp32
    tb = D.tables(D.linear_betas(1000))
    tab = gd.device_table(x.device).cpu().numpy()
    assert np.array_equal(tab[6], tb["posterior_log_variance_clipped"].astype(np.float32))
    assert np.array_equal(tab[0], tb["sqrt_alphas_cumprod"].astype(np.float32))


def test_masked_mse_and_grad():
    B, T, Fd = 3, 20, 150
    pred, tgt = rnd(B, T, Fd), rnd(B, T, Fd, seed=1)
    lens = torch.tensor([20, 7, 1])
    pd = pred.double().requires_grad_(True)
    mask = (torch.arange(T)[None] < lens[:, None]).double()
    loss_ref = (((pd - tgt.double()) ** 2).mean(-1) * mask).sum() / mask.sum()
    loss_ref.backward()
    loss, dp = torch.zeros(1, device=DEV), torch.zeros(B, T, Fd, device=DEV)
    scr = torch.zeros(_lib.NORM_BLOCKS, device=DEV)
    _lib.check(_lib.lib().hig_masked_mse(P(pred.to(DEV)), P(tgt.to(DEV)), P(lens.to(DEV)), B, T, Fd, P(loss),
                                         P(dp), P(scr), _lib.stream_ptr()))
    assert abs(loss.item() - loss_ref.item()) < 1e-5 * abs(loss_ref.item())
    assert rel(dp, pd.grad) < 1e-6


def test_clip_adam_matches_torch():
    n = 100003
    p0, g = rnd(n), rnd(n, seed=1, scale=0.01)
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.Adam([ref], lr=2e-4)
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    scr, gn = torch.zeros(_lib.NORM_BLOCKS, device=DEV), torch.zeros(1, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    L = _lib.lib()
    for it in range(3):
        gi = g * (it + 1) * (40.0 if it == 1 else 1.0)   # step 1 is above the 0.5 clip threshold
        ref.grad = gi.double() / 2                        # world = 2: mean of the summed gradient
        tot = torch.nn.utils.clip_grad_norm_([ref], 0.5)
        opt.step()
        gg = gi.to(DEV)
        _lib.check(L.hig_sumsq_partial(P(gg), n, 0.5, P(scr), _lib.stream_ptr()))
        _lib.check(L.hig_clip_adam(P(p), P(gg), P(m), P(v), n, 2e-4, 0.9, 0.999, 1e-8, 0.5, 0.5, P(scr), P(gn),
                                   P(step), _lib.stream_ptr()))
        assert abs(gn.item() - tot.item()) < 1e-5 * tot.item()
        assert rel(p, ref.data) < 1e-6
    assert step.item() == 3


def test_transpose_batch_matches_torch():
    """Several W -> W^T copies in one launch (ragged extents, up to 12 matrices)."""
    import ctypes as C
    shapes = [(512, 512), (1024, 512), (150, 512), (70, 33), (1, 5), (1536, 512), (64, 64)]
    src = [torch.randn(r, c, device=DEV) for r, c in shapes]
    dst = [torch.full((c, r), float("nan"), device=DEV) for r, c in shapes]
    n = len(shapes)
    srcs = (C.c_void_p * n)(*[t.data_ptr() for t in src])
    dsts = (C.c_void_p * n)(*[t.data_ptr() for t in dst])
    rows = (C.c_int32 * n)(*[r for r, _ in shapes])
    cols = (C.c_int32 * n)(*[c for _, c in shapes])
    _lib.check(_lib.lib().hig_transpose_batch(n, srcs, dsts, rows, cols, _lib.stream_ptr()))
    for a, b in zip(src, dst):
        assert torch.equal(b, a.t())
    with pytest.raises(RuntimeError, match="matrices"):
        _lib.check(_lib.lib().hig_transpose_batch(13, srcs, dsts, rows, cols, _lib.stream_ptr()))


@pytest.mark.parametrize("I,J,R", [(512, 512, 1000), (1536, 512, 196), (64, 32, 77), (132, 256, 640)])
def test_gemm_wgrad_with_fused_column_sums(I, J, R):
    """wgrad layout (X = dC reduce-slow, Y = activations reduce-slow) with xcolsum: C = X^T-contracted product
    and xcolsum[i] = sum_r X[r][i] (the bias gradient) from the same pass."""
    import ctypes as C
    X, Y = rnd(R, I, seed=3), rnd(R, J, seed=4)
    out = torch.full((I, J), float("nan"), device=DEV)
    xs = torch.full((I,), float("nan"), device=DEV)
    d = _lib.GemmDesc()
    Xd, Yd = X.to(DEV), Y.to(DEV)
    d.X, d.ldx, d.x_rs, d.Y, d.ldy, d.y_rs = Xd.data_ptr(), I, 1, Yd.data_ptr(), J, 1
    d.C, d.ldc, d.I, d.J, d.R = out.data_ptr(), J, I, J, R
    d.xf, d.epi, d.prec = _lib.XF_NONE, _lib.EPI_NONE, _lib.PREC_F32
    d.xcolsum = xs.data_ptr()
    _lib.check(_lib.lib().hig_gemm(C.byref(d), _lib.stream_ptr()))
    assert rel(out, X.double().t() @ Y.double()) < 2e-6
    assert rel(xs, X.double().sum(0)) < 2e-6
    d.x_rs, d.ldx = 0, R                     # a reduce-contiguous X cannot provide the sums: loud error
    with pytest.raises(RuntimeError, match="xcolsum"):
        _lib.check(_lib.lib().hig_gemm(C.byref(d), _lib.stream_ptr()))


@pytest.mark.parametrize("B,T,H", [(3, 128, 8), (2, 131, 4), (64, 196, 8), (2, 257, 8), (5, 300, 2)])
def test_wave_autonomous_apply_kernel_fp32_hd64(B, T, H):
    """hig_linattn_apply, exact fp32, head dim 64, >= 128 rows: the wave-autonomous kernel (16-row tiles per wave, MFMA operand
    roles chosen so that neither q nor y is transposed, transformer.py:111,116-118) against the fp64 definition -- query rows
    inside the stacked q/k/v buffer (row stride 3 d) and on their own (cross-attention, stride d), ragged last tile, nothing
    written outside the (B T, d) output."""
    hd = 64
    d = H * hd
    L = _lib.lib()
    s = _lib.stream_ptr()
    A = (rnd(B, H, hd, hd, seed=3) * 0.2).to(DEV)
    for ldq in (3 * d, d):
        qb = rnd(B * T, ldq, scale=1.5, seed=ldq).to(DEV)
        q = qb[:, :d]
        ybig = torch.full((B * T + 3, d + 8), 5.0, device=DEV)
        y = ybig[:B * T, :d]
        _lib.check(L.hig_linattn_apply(P(q), ldq, P(A), P(y), d + 8, B, T, H, hd, s))
        torch.cuda.synchronize()
        p = torch.softmax(q.double().view(B, T, H, hd), dim=-1)
        ref = torch.einsum("bnhd,bhdl->bnhl", p, A.double()).reshape(B * T, d)
        assert rel(y, ref) < 2e-6, (ldq, rel(y, ref))
        assert (y.double() - ref).abs().max().item() < 2e-5
        assert bool((ybig[:, d:] == 5.0).all()) and bool((ybig[B * T:] == 5.0).all())
        y2 = torch.empty(B * T, d, device=DEV)
        _lib.check(L.hig_linattn_apply(P(q), ldq, P(A), P(y2), d, B, T, H, hd, s))
        torch.cuda.synchronize()
        assert torch.equal(y2, y.contiguous())
