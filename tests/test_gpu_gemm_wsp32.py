"""Exact-fp32 weight-stationary GEMM with specialised waves (csrc/gemm_wsp32.hip) through hig_gemm: every epilogue it
serves against the fp64 product, at the model's shapes (K = 512 / 1024, many rows), ragged row counts, the LayerNorm-fold
producer -> consumer pair against F.layer_norm, bitwise repeatability and independence of the batch split.
hig_gemm_wsp32_launches() proves that the launch went to this kernel and not to the tiled one."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import hig_amd  # noqa: E402,F401
from hig_amd import _lib  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def launch(X, W, epi=0, bias=None, res=None, aux=None, stats_out=None, stats_in=None, colsum=None, out=None, expect_wsp=True):
    I, R = X.shape
    J = W.shape[0]
    out = torch.full((I, J), float("nan"), device=DEV) if out is None else out
    d = _lib.GemmDesc()
    d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc = X.data_ptr(), X.stride(0), W.data_ptr(), W.stride(0), out.data_ptr(), out.stride(0)
    d.I, d.J, d.R, d.epi = I, J, R, epi
    if bias is not None:
        d.bias = bias.data_ptr()
    if res is not None:
        d.res, d.ldr = res.data_ptr(), res.stride(0)
    if aux is not None:
        d.aux, d.ldaux = aux.data_ptr(), aux.stride(0)
    if stats_out is not None:
        d.row_stats_out = stats_out.data_ptr()
    if stats_in is not None:
        d.row_stats_in, d.ln_colsum = stats_in.data_ptr(), colsum.data_ptr()
    L = _lib.lib()
    before = L.hig_gemm_wsp32_launches()
    _lib.check(L.hig_gemm(C.byref(d), _lib.stream_ptr()))
    torch.cuda.synchronize()
    ran = L.hig_gemm_wsp32_launches() - before == 1
    assert ran == expect_wsp, "launch %s the weight-stationary fp32 kernel" % ("did not reach" if expect_wsp else "reached")
    return out


CASES = [(12544, 512, 512), (12544, 1024, 512), (12544, 1536, 512), (12544, 512, 1024), (6272, 512, 512), (6272, 1536, 512),
         (12539, 512, 512), (2050, 1024, 512), (5824, 1536, 512), (5824, 512, 1024), (2051, 512, 1024), (4096, 1792, 512),
         (9600, 1024, 1024)]


@pytest.mark.parametrize("I,J,R", CASES)
def test_every_epilogue_against_fp64(I, J, R):
    X, W, b, r = rnd(I, R).to(DEV), rnd(J, R, seed=1, scale=0.05).to(DEV), rnd(J, seed=2).to(DEV), rnd(I, J, seed=3).to(DEV)
    prod = X.double() @ W.double().T
    assert rel(launch(X, W), prod) < 2e-6
    assert rel(launch(X, W, epi=_lib.EPI_BIAS, bias=b), prod + b.double()) < 2e-6
    assert rel(launch(X, W, epi=_lib.EPI_BIAS_RES, bias=b, res=r), prod + b.double() + r.double()) < 2e-6
    assert rel(launch(X, W, epi=_lib.EPI_RES, res=r), prod + r.double()) < 2e-6
    z = torch.full((I, J), float("nan"), device=DEV)
    o = launch(X, W, epi=_lib.EPI_BIAS_GELU, bias=b, aux=z)
    assert rel(o, F.gelu(prod + b.double())) < 2e-6
    assert rel(z, prod + b.double()) < 2e-6
    assert torch.equal(launch(X, W, epi=_lib.EPI_BIAS_GELU, bias=b), o)          # with and without the second output
    zz = r.double()
    dg = 0.5 * (1 + torch.erf(zz / 2 ** 0.5)) + zz * torch.exp(-0.5 * zz * zz) / (2 * torch.pi) ** 0.5
    assert rel(launch(X, W, epi=_lib.EPI_DGELU, aux=r), prod * dg) < 2e-6


@pytest.mark.parametrize("I,J,R", [(12544, 512, 512), (6272, 1536, 512), (5824, 512, 1024)])
def test_exact_on_small_integers_and_bitwise_repeatable(I, J, R):
    g = torch.Generator().manual_seed(5)
    Xi = torch.randint(-4, 5, (I, R), generator=g).float()
    Wi = torch.randint(-4, 5, (J, R), generator=g).float()
    assert torch.equal(launch(Xi.to(DEV), Wi.to(DEV)).cpu(), Xi @ Wi.T)
    X, W, b = rnd(I, R).to(DEV), rnd(J, R, seed=1).to(DEV), rnd(J, seed=2).to(DEV)
    a = launch(X, W, epi=_lib.EPI_BIAS, bias=b)
    for _ in range(5):
        assert torch.equal(launch(X, W, epi=_lib.EPI_BIAS, bias=b), a)
    # a row's result does not depend on which rows travel with it (no tile schedule in the sum order): the two halves of the
    # batch computed alone equal the whole batch bit for bit
    h = (I // 2) // 16 * 16 + 3
    assert torch.equal(launch(X[:h], W, epi=_lib.EPI_BIAS, bias=b), a[:h])
    assert torch.equal(launch(X[h:], W, epi=_lib.EPI_BIAS, bias=b), a[h:])


def test_strided_operands_and_row_padding_untouched():
    """ldx / ldc / ldr larger than the extents (the q/k/v buffer is one (M, 3 d) matrix): nothing outside [I][J] is written."""
    I, J, R = 6272, 512, 512
    Xb, W, b = rnd(I, 3 * R).to(DEV), rnd(J, R, seed=1, scale=0.05).to(DEV), rnd(J, seed=2).to(DEV)
    Cb = torch.full((I + 4, 3 * J), 7.0, device=DEV)
    rb = rnd(I, 2 * J, seed=4).to(DEV)
    X, out, r = Xb[:, R:2 * R], Cb[:I, J:2 * J], rb[:, J:]
    launch(X, W, epi=_lib.EPI_BIAS_RES, bias=b, res=r, out=out)
    assert rel(out, X.double() @ W.double().T + b.double() + r.double()) < 2e-6
    assert bool((Cb[:I, :J] == 7.0).all()) and bool((Cb[:I, 2 * J:] == 7.0).all()) and bool((Cb[I:] == 7.0).all())


@pytest.mark.parametrize("I", [12544, 6277])
def test_layernorm_fold_producer_and_consumer(I):
    """row_stats_out (EPI_BIAS_RES) -> row_stats_in + ln_colsum (EPI_BIAS): LayerNorm(h) W^T + b without a LayerNorm pass
    (hig_gemm_desc, include/hig.h); rows with a mean hundreds of times their spread keep their variance."""
    d = 512
    a, Wo, bo, hin = rnd(I, d).to(DEV), rnd(d, d, seed=1, scale=0.05).to(DEV), rnd(d, seed=2).to(DEV), rnd(I, d, seed=3).to(DEV)
    hin[: I // 2] += 300.0
    stats = torch.full((I, d // 64, 2), float("nan"), device=DEV)
    h = launch(a, Wo, epi=_lib.EPI_BIAS_RES, bias=bo, res=hin, stats_out=stats)
    href = a.double() @ Wo.double().T + bo.double() + hin.double()
    assert rel(h, href) < 2e-6
    hp = h.double().view(I, d // 64, 64)
    assert rel(stats[:, :, 0], hp.sum(-1)) < 1e-6
    assert rel(stats[:, :, 1], ((hp - hp.mean(-1, keepdim=True)) ** 2).sum(-1)) < 1e-5
    gamma, beta = (1 + 0.1 * rnd(d, seed=5)).to(DEV), (0.1 * rnd(d, seed=6)).to(DEV)
    for J in (512, 1536):
        W, b = rnd(J, d, seed=7, scale=0.05).to(DEV), rnd(J, seed=8).to(DEV)
        Wp = (W * gamma).contiguous()
        colsum = Wp.double().sum(-1).float()
        bp = (b.double() + W.double() @ beta.double()).float()
        out = launch(h, Wp, epi=_lib.EPI_BIAS, bias=bp, stats_in=stats, colsum=colsum)
        ref = F.linear(F.layer_norm(h.double(), (d,), gamma.double(), beta.double()), W.double(), b.double())
        assert rel(out, ref) < 3e-5        # (the un-normalised product loses log2(|mean| / spread) bits: same bar as the tiled kernel's fold)
        assert rel(out[I // 2:], ref[I // 2:]) < 3e-6


def test_shapes_it_declines_still_run_on_the_tiled_kernel():
    for I, J, R in [(1024, 512, 512), (6272, 512, 384), (6272, 520, 512), (6272, 320, 256)]:   # few rows, K not 256 / 512 / 1024, J not in whole panels
        X, W, b = rnd(I, R).to(DEV), rnd(J, R, seed=1, scale=0.05).to(DEV), rnd(J, seed=2).to(DEV)
        assert rel(launch(X, W, epi=_lib.EPI_BIAS, bias=b, expect_wsp=False), X.double() @ W.double().T + b.double()) < 2e-6


@pytest.mark.parametrize("I,J,R", [(12544, 512, 1536), (6272, 512, 2048), (2060, 1536, 1536)])
def test_reduce_ranges_beyond_the_weight_panel_go_through_in_two_passes(I, J, R):
    """K = 1536 / 2048 (data gradient of the stacked q/k/v projection): K = 1024 first, the rest added with C as its own residual."""
    X, W, b, r = rnd(I, R).to(DEV), rnd(J, R, seed=1, scale=0.05).to(DEV), rnd(J, seed=2).to(DEV), rnd(I, J, seed=3).to(DEV)
    prod = X.double() @ W.double().T
    L = _lib.lib()
    for epi, kw, ref in [(0, {}, prod), (_lib.EPI_BIAS, dict(bias=b), prod + b.double()), (_lib.EPI_RES, dict(res=r), prod + r.double()),
                         (_lib.EPI_BIAS_RES, dict(bias=b, res=r), prod + b.double() + r.double())]:
        out = torch.full((I, J), float("nan"), device=DEV)
        d = _lib.GemmDesc()
        d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc, d.I, d.J, d.R, d.epi = X.data_ptr(), R, W.data_ptr(), R, out.data_ptr(), J, I, J, R, epi
        if "bias" in kw:
            d.bias = b.data_ptr()
        if "res" in kw:
            d.res, d.ldr = r.data_ptr(), J
        before = L.hig_gemm_wsp32_launches()
        _lib.check(L.hig_gemm(C.byref(d), _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert L.hig_gemm_wsp32_launches() - before == 2
        assert rel(out, ref) < 2e-6


@pytest.mark.parametrize("I,J,R,splits", [(512, 512, 12544, 16), (1536, 512, 12544, 5), (512, 1024, 6272, 8), (1024, 512, 12539, 8),
                                          (128, 256, 4099, 0), (512, 512, 2048, 0)])
def test_weight_gradient_kernel_against_fp64(I, J, R, splits):
    """wgrad_wsp32.hip through hig_gemm_split (the weight-gradient form of the fp32 backward: dW = dC^T . act over R rows in
    `splits` slabs + the deterministic slab reduction, dbias = column sums of dC from the same pass): fp64 product, exact on
    small integers, bitwise repeatable, row counts that are not a multiple of the 32-row stage."""
    L = _lib.lib()
    dC, act = rnd(R, I, seed=3).to(DEV), rnd(R, J, seed=4).to(DEV)
    d = _lib.GemmDesc()
    out, xs = torch.full((I, J), float("nan"), device=DEV), torch.full((I,), float("nan"), device=DEV)
    d.X, d.ldx, d.x_rs, d.Y, d.ldy, d.y_rs = dC.data_ptr(), I, 1, act.data_ptr(), J, 1
    d.C, d.ldc, d.I, d.J, d.R, d.xcolsum = out.data_ptr(), J, I, J, R, xs.data_ptr()
    n = L.hig_gemm_split_scratch_floats(C.byref(d), splits)
    slabs = torch.full((n,), float("nan"), device=DEV)

    def run():
        _lib.check(L.hig_gemm_split(C.byref(d), splits, slabs.data_ptr(), n, _lib.stream_ptr()))
        torch.cuda.synchronize()
        return out.clone(), xs.clone()
    o, x = run()
    assert rel(o, dC.double().t() @ act.double()) < 2e-6
    assert rel(x, dC.double().sum(0)) < 2e-6
    for _ in range(3):
        slabs.fill_(float("nan"))
        o2, x2 = run()
        assert torch.equal(o2, o) and torch.equal(x2, x)
    g = torch.Generator().manual_seed(5)
    Ci, Ai = torch.randint(-3, 4, (R, I), generator=g).float(), torch.randint(-3, 4, (R, J), generator=g).float()
    dC.copy_(Ci)
    act.copy_(Ai)
    o, x = run()
    assert torch.equal(o.cpu(), Ci.t() @ Ai) and torch.equal(x.cpu(), Ci.sum(0))


@pytest.mark.parametrize("I,J", [(4928, 8192), (4928, 1024), (2464, 4096), (4925, 384), (2050, 128)])
def test_k256_instance_plain_and_bias(I, J):
    """K = 256 (text_latent_dim): the cross-attention key/value projection of the text rows (transformer.py:146,150), all layers'
    weights stacked in ONE launch (J = 2 d L) or one layer's (J = 2 d); 128-column panels, one K part per matrix wave, two
    epilogue passes per service wave.  Other epilogues are declined at this K (the tiled kernel serves them)."""
    R = 256
    X, W, b = rnd(I, R).to(DEV), rnd(J, R, seed=1, scale=0.05).to(DEV), rnd(J, seed=2).to(DEV)
    prod = X.double() @ W.double().T
    assert rel(launch(X, W), prod) < 2e-6
    o = launch(X, W, epi=_lib.EPI_BIAS, bias=b)
    assert rel(o, prod + b.double()) < 2e-6
    assert (o.double().cpu() - (prod + b.double()).cpu()).abs().max().item() < 1e-4
    for _ in range(3):
        assert torch.equal(launch(X, W, epi=_lib.EPI_BIAS, bias=b), o)
    h = (I // 2) // 16 * 16 + 5
    if h >= 2048:
        assert torch.equal(launch(X[:h], W, epi=_lib.EPI_BIAS, bias=b), o[:h])
    g = torch.Generator().manual_seed(6)
    Xi, Wi = torch.randint(-4, 5, (I, R), generator=g).float(), torch.randint(-4, 5, (J, R), generator=g).float()
    assert torch.equal(launch(Xi.to(DEV), Wi.to(DEV)).cpu(), Xi @ Wi.T)
    r = rnd(I, J, seed=3).to(DEV)
    assert rel(launch(X, W, epi=_lib.EPI_BIAS_RES, bias=b, res=r, expect_wsp=False), prod + b.double() + r.double()) < 2e-6
    # strided output (one layer's slice of the stacked key/value buffer)
    if J == 1024:
        big = torch.full((I, 4 * J), 3.0, device=DEV)
        launch(X, W, epi=_lib.EPI_BIAS, bias=b, out=big[:, J:2 * J])
        assert torch.equal(big[:, J:2 * J], o) and bool((big[:, :J] == 3.0).all()) and bool((big[:, 2 * J:] == 3.0).all())
