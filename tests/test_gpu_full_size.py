"""Parity at BASELINE.json's FULL sizes on the MI355X (configs 2 and 3): the backward at B=64, T=196, d=512, L=8
against the CPU oracle's autograd, the weight-gradient GEMM at the product's own split-R geometry, and the
hipGraph-captured 1000-step sampling loop at B=32 against the eager loop.  Tolerances are written where they apply;
the gate is north_star's 1e-3 relative fp32."""
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
CONFIG2 = dict(fill.CASES["width"], B=64, lengths=tuple([196] * 40 + list(range(60, 180, 5))),
               t=tuple(int(v) for v in np.linspace(0, 999, 64)))


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def build(c, **kw):
    m = hig_amd.MotionTransformer(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"],
                                  ff_size=c["ff"], num_layers=c["L"], num_heads=c["H"],
                                  text_latent_dim=c["Lt"], **kw)
    m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
    return m.to(DEV)


CONFIG5 = dict(B=32, T=300, F=150, d=1024, H=8, L=12, ff=1024, N=77, Lt=256, num_frames=300,
               lengths=tuple([300, 211] + [300] * 20 + list(range(30, 300, 27))), t=tuple(int(v) for v in np.linspace(0, 999, 32)))


@pytest.mark.parametrize("B", [32, 64])
def test_fp32_per_call_forward_at_config2_size_equals_the_cached_form(B):
    """hig_denoiser_fwd_text at the size where BOTH the frame-row GEMMs and the forked text-side key/value GEMMs (B x 77 rows: 624
    tiles of 64 x 64 at B = 32) take the split tail of the fp32 GEMM: each side has its own ticket / partial-sum scratch (ADVICE r04:
    shared, a workgroup could draw another launch's "last" ticket).  The cached context is built WITHOUT a split tail (another
    summation order of the last round's tiles), so the two forms agree to rounding, not bitwise; the per-call form must repeat bit
    for bit -- tickets or partial sums shared between the two concurrent launches would show as run-to-run differences."""
    c = dict(CONFIG2, B=B, lengths=CONFIG2["lengths"][:B], t=CONFIG2["t"][:B])
    m = build(c).eval()
    gi = {k: v.to(DEV) for k, v in fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"]).items()}

    def fwd():
        with torch.no_grad():
            return m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])

    m.cache_text_context = True
    ref = fwd()
    assert torch.isfinite(ref).all()
    m.cache_text_context = False
    first = fwd()
    assert rel(first, ref) < 2e-6
    for _ in range(12):
        assert torch.equal(fwd(), first)


@pytest.mark.parametrize("no_eff", [False, True])
def test_config5_as_specified_bf16_storage_forward_slice_against_oracle(no_eff):
    """BASELINE config 5 AS SPECIFIED -- B=32, T=300, d=1024, L=12, head dim 128, bf16 storage -- with linear attention and with
    no_eff=True (full softmax attention): the whole batch runs on the GPU (9 600 rows: the K = 1024 variants of the
    weight-stationary GEMM at their full-size work split, `apply_sty16<128>` / `ctx16<128>` at 32 samples, the LayerNorm fold at
    K = 1024); samples are independent, so the first two samples (one full-length, one ragged) are compared with the fp32 CPU
    oracle run on just those two (transformer.py:60-194 at latent_dim=1024).  Gates: finite everywhere, rel-L2 of the slice
    at the bf16 level (3e-2) and clearly above fp32 noise."""
    c = CONFIG5
    assert len(c["lengths"]) == c["B"]
    m = build(c, storage="bf16", no_eff=no_eff).eval()
    inp = fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"])
    gi = {k: v.to(DEV) for k, v in inp.items()}
    with torch.no_grad():
        out = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
        n = 2
        p = fill.core_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"])
        ref = R.denoiser_forward(p, inp["x"][:n], inp["t"][:n], inp["length"][:n], inp["xf_proj"][:n], inp["xf_out"][:n], c["H"], c["L"],
                                 no_eff=no_eff)
    assert out.shape == (c["B"], c["T"], c["F"]) and torch.isfinite(out).all()
    errs = [rel(out[b, :min(c["lengths"][b], c["T"])], ref[b, :min(c["lengths"][b], c["T"])]) for b in range(n)]
    print("config 5 (B=32 T=300 d=1024 L=12, bf16 storage, no_eff=%s): rel-L2 of the valid rows of samples 0, 1 vs the fp32 oracle %s" %
          (no_eff, ["%.2e" % e for e in errs]))
    assert all(1e-4 < e < 3e-2 for e in errs), errs


_ORACLE_BWD = {}


def _config2_oracle_backward():
    """torch autograd of the fp32 CPU oracle at BASELINE config 2 (B=64, T=196, d=512, L=8), loss = masked MSE against the fixed
    target: ~15 s of host time, computed ONCE for the fp32 and the bf16-storage test below (same inputs, same target)."""
    if "v" not in _ORACLE_BWD:
        c = CONFIG2
        inp = fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"])
        target = fill.tensor_for("full.target", inp["x"].shape) * 10.0
        names = fill.core_param_shapes(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"])
        p = {k: v.clone().requires_grad_(True) for k, v in
             fill.core_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"]).items()}
        xin = tuple(inp[k].clone().requires_grad_(True) for k in ("x", "xf_proj", "xf_out"))
        ref = R.denoiser_forward(p, xin[0], inp["t"], inp["length"], xin[1], xin[2], c["H"], c["L"])
        ref_loss = D.masked_mse(ref, target, R.src_mask(c["T"], inp["length"]))
        ref_loss.backward()
        _ORACLE_BWD["v"] = (names, p, xin, ref.detach(), ref_loss.detach())
    return _ORACLE_BWD["v"]


def test_config2_size_bf16_storage_backward_every_gradient_against_oracle_autograd():
    """The bf16-storage training forward + backward at BASELINE config 2 itself (B=64, T=196, d=512, L=8: 12 544 rows -- the
    row count at which the weight-gradient kernel runs sixteen row slices per tile and the weight-stationary GEMMs their
    full-size work split), EVERY parameter gradient (329 tensors at L = 8) and the three input gradients against torch autograd of the fp32 CPU
    oracle.  Same gates as tests/test_gpu_bf16_training.py (which stops at 3 136 rows): loss 1e-2, global gradient norm 2e-2,
    direction of the whole gradient 1 - 1e-3, median tensor 3e-2, no tensor beyond 0.15 (trainers/ddpm_trainer.py:172-187)."""
    c = CONFIG2
    m = build(c, storage="bf16").train()
    inp = fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"])
    gi = {k: v.to(DEV) for k, v in inp.items()}
    target = fill.tensor_for("full.target", inp["x"].shape) * 10.0
    x, xp, xo = (gi[k].clone().requires_grad_(True) for k in ("x", "xf_proj", "xf_out"))
    out = m(x, gi["t"], length=gi["length"], xf_proj=xp, xf_out=xo)
    mask = m.generate_src_mask(c["T"], gi["length"]).to(DEV)
    loss = (((out - target.to(DEV)) ** 2).mean(-1) * mask).sum() / mask.sum()
    loss.backward()
    names, p, (xr, xpr, xor_), ref, ref_loss = _config2_oracle_backward()
    named = dict(m.named_parameters())
    e_out = rel(out, ref)
    assert 1e-4 < e_out < 3e-2, e_out                       # the bf16 path really ran, at the bf16 level
    assert abs(loss.item() - ref_loss.item()) < 1e-2 * abs(ref_loss.item())
    zero_grad = [k for k in names if k.endswith(".key.bias")]          # exactly zero in the reference (softmax over tokens)
    live = [k for k in names if k not in zero_grad]
    for k in live:
        assert named[k].grad is not None and torch.isfinite(named[k].grad).all(), k
    gn = torch.sqrt(sum((named[k].grad.double() ** 2).sum() for k in live)).item()
    gn_ref = torch.sqrt(sum((p[k].grad.double() ** 2).sum() for k in live)).item()
    dot = sum((named[k].grad.double().cpu() * p[k].grad.double()).sum() for k in live).item()
    errs = sorted(((rel(named[k].grad, p[k].grad), k) for k in live), reverse=True)
    print("bf16-storage backward at config 2: out %.2e, loss %.6f vs %.6f, grad norm %.6e vs %.6e (rel %.2e), cosine %.6f" %
          (e_out, loss.item(), ref_loss.item(), gn, gn_ref, abs(gn - gn_ref) / gn_ref, dot / (gn * gn_ref)))
    print("   worst tensors: " + ", ".join("%s %.2e" % (k, e) for e, k in errs[:5]))
    print("   median rel-L2 over %d tensors: %.2e" % (len(errs), errs[len(errs) // 2][0]))
    assert abs(gn - gn_ref) < 2e-2 * gn_ref, (gn, gn_ref)
    assert dot / (gn * gn_ref) > 1 - 1e-3
    assert errs[0][0] < 0.15, errs[0]                       # no tensor is garbage / unwritten
    assert errs[len(errs) // 2][0] < 3e-2
    for a, b in ((x.grad, xr.grad), (xp.grad, xpr.grad), (xo.grad, xor_.grad)):
        assert rel(a, b) < 5e-2
    for k in zero_grad:
        assert named[k].grad.norm().item() < 2e-2 * gn_ref, k


def test_config2_size_backward_against_oracle_autograd():
    """B=64, T=196, d=512, L=8, ragged lengths: loss = masked MSE against a fixed target (the training loss,
    ddpm_trainer.py:172-178); every gradient the product computes at this size comes from 16-32 split-R slabs on two
    streams.  Checked against torch autograd of the fp32 CPU oracle: global gradient norm, input gradients and named
    parameter gradients from every kind of kernel (row-kernel LayerNorm grads, few-row emb GEMMs, split-R wgrads)."""
    c = CONFIG2
    m = build(c).train()
    inp = fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"])
    gi = {k: v.to(DEV) for k, v in inp.items()}
    target = fill.tensor_for("full.target", inp["x"].shape) * 10.0
    x, xp, xo = (gi[k].clone().requires_grad_(True) for k in ("x", "xf_proj", "xf_out"))
    out = m(x, gi["t"], length=gi["length"], xf_proj=xp, xf_out=xo)
    mask = m.generate_src_mask(c["T"], gi["length"]).to(DEV)
    loss = (((out - target.to(DEV)) ** 2).mean(-1) * mask).sum() / mask.sum()
    loss.backward()

    names, p, (xr, xpr, xor_), ref, ref_loss = _config2_oracle_backward()

    assert rel(out, ref) < 5e-5
    assert abs(loss.item() - ref_loss.item()) < 1e-5 * abs(ref_loss.item())
    assert rel(x.grad, xr.grad) < 1e-3 and rel(xp.grad, xpr.grad) < 1e-3 and rel(xo.grad, xor_.grad) < 1e-3
    named = dict(m.named_parameters())
    gn = torch.sqrt(sum((named[k].grad.double() ** 2).sum() for k in names)).item()
    gn_ref = torch.sqrt(sum((p[k].grad.double() ** 2).sum() for k in names)).item()
    assert abs(gn - gn_ref) < 2e-4 * gn_ref, (gn, gn_ref)
    checked = ("temporal_decoder_blocks.0.sa_block.proj_out.emb_layers.1.weight",    # few-row GEMM, 200 of 347 MB
               "temporal_decoder_blocks.7.ffn.proj_out.emb_layers.1.bias",
               "temporal_decoder_blocks.3.sa_block.key.weight",                       # split-R wgrad, fused q/k/v
               "temporal_decoder_blocks.5.ca_block.value.weight",                     # text-side wgrad (M = B*77)
               "temporal_decoder_blocks.2.ffn.linear1.weight", "temporal_decoder_blocks.6.ffn.linear2.weight",
               "temporal_decoder_blocks.4.ca_block.proj_out.out_layers.2.weight",
               "temporal_decoder_blocks.1.sa_block.norm.weight", "temporal_decoder_blocks.7.ca_block.text_norm.bias",
               "temporal_decoder_blocks.0.ffn.linear1.bias", "time_embed.2.weight", "joint_embed.weight",
               "sequence_embedding", "out.weight", "out.bias")
    worst = max((rel(named[k].grad, p[k].grad), k) for k in checked)
    assert worst[0] < 1e-3, worst                     # the north-star gate
    assert worst[0] < 3e-4, worst                     # what fp32 accumulation over 12 544 rows delivers
    # every parameter gradient, loosely (catches a tensor that was never written).  The key biases have an exactly
    # zero gradient (a per-channel constant drops out of the softmax over tokens, transformer.py:112,148): both sides
    # hold rounding noise there, so they are held to "tiny" instead of to each other
    zero_grad = [k for k in names if k.endswith(".key.bias")]
    for k in zero_grad:
        assert named[k].grad.norm().item() < 1e-5 * gn_ref and p[k].grad.norm().item() < 1e-5 * gn_ref, k
    all_worst = max((rel(named[k].grad, p[k].grad), k) for k in names if k not in zero_grad)
    assert all_worst[0] < 2e-3, all_worst


@pytest.mark.parametrize("I,J,R,splits", [(1024, 512, 12544, 0), (1536, 512, 12544, 0), (512, 512, 12544, 7),
                                          (512, 256, 4928, 0)])
def test_wgrad_split_geometry_of_the_product(I, J, R, splits):
    """The weight-gradient GEMM exactly as hig_denoiser_bwd launches it at config 2: dW (I x J) = dC^T . act over
    R = 12 544 rows (R = 4 928 for the text side), reduce range split into slabs by the library's own rule (or a forced
    odd count), slab reduction, bias gradient (column sums of dC) from the same pass.  fp64 reference."""
    g = torch.Generator().manual_seed(I + J + R)
    X = torch.randn(R, I, generator=g)
    Y = torch.randn(R, J, generator=g)
    Xd, Yd = X.to(DEV), Y.to(DEV)
    out = torch.full((I, J), float("nan"), device=DEV)
    xs = torch.full((I,), float("nan"), device=DEV)
    d = _lib.GemmDesc()
    d.X, d.ldx, d.x_rs, d.Y, d.ldy, d.y_rs = Xd.data_ptr(), I, 1, Yd.data_ptr(), J, 1
    d.C, d.ldc, d.I, d.J, d.R = out.data_ptr(), J, I, J, R
    d.xf, d.epi, d.prec = _lib.XF_NONE, _lib.EPI_NONE, _lib.PREC_F32
    d.xcolsum = xs.data_ptr()
    L = _lib.lib()
    n = L.hig_gemm_split_scratch_floats(C.byref(d), splits)
    assert n > 0
    slabs = torch.full((n,), float("nan"), device=DEV)
    _lib.check(L.hig_gemm_split(C.byref(d), splits, _lib.ptr(slabs), n, _lib.stream_ptr()))
    ref = X.double().t() @ Y.double()
    assert rel(out, ref) < 3e-6
    assert ((out.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 2e-5
    assert rel(xs, X.double().sum(0)) < 3e-6
    first = out.clone()
    _lib.check(L.hig_gemm_split(C.byref(d), splits, _lib.ptr(slabs), n, _lib.stream_ptr()))
    assert torch.equal(first, out)                     # fixed-order slab reduction: bitwise reproducible
    with pytest.raises(RuntimeError, match="scratch"):
        _lib.check(L.hig_gemm_split(C.byref(d), 4, _lib.ptr(slabs), 16, _lib.stream_ptr()))


def _diffusion(n):
    return hig_amd.GaussianDiffusion(betas=gdm.get_named_beta_schedule("linear", n),
                                     model_mean_type=gdm.ModelMeanType.EPSILON,
                                     model_var_type=gdm.ModelVarType.FIXED_SMALL, loss_type=gdm.LossType.MSE)


def _zero_noise_loop(m, gd, use_graph, x0, kw):
    gd.use_hip_graph, gd._debug_zero_noise = use_graph, True
    old = gdm.th
    if not use_graph:
        proxy = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
        proxy.randn_like = lambda x, **_: torch.zeros_like(x)
        gdm.th = proxy
    try:
        return gd.p_sample_loop(m, x0.shape, noise=x0.clone(), clip_denoised=False, model_kwargs=kw)
    finally:
        gdm.th = old


@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_config3_captured_1000_step_loop_equals_eager_loop(mode):
    """BASELINE config 3: the 1000-step p_sample_loop at B=32, T=196 on the config-2 model, captured as ONE hipGraph
    replayed 1000 times, against the eager loop (one launch sequence per step, fresh `t` tensor from the host like
    gaussian_diffusion.py:718-769) with the noise draws zeroed on both sides: same kernels, same order, so the two
    must agree to rounding, and stay finite over the whole chain."""
    c = dict(fill.CASES["width"], B=32)
    m = build(c, precision=mode).eval()
    with torch.no_grad():       # a random eps-network fed back 1000 times: keep its gain small so the chain stays in range
        m.out.weight.mul_(0.05)
        m.out.bias.mul_(0.05)
    g = torch.Generator().manual_seed(11)
    kw = {"xf_proj": torch.randn(32, 4 * c["d"], generator=g).to(DEV),
          "xf_out": torch.randn(32, c["N"], c["Lt"], generator=g).to(DEV),
          "length": torch.tensor([196] * 20 + list(range(100, 196, 8)), device=DEV)}
    x0 = torch.randn(32, 196, c["F"], generator=g).to(DEV)
    eager = _zero_noise_loop(m, _diffusion(1000), False, x0, kw)
    graph = _zero_noise_loop(m, _diffusion(1000), True, x0, kw)
    assert torch.isfinite(graph).all() and torch.isfinite(eager).all()
    assert rel(graph, eager) < 1e-5
    # with real noise every replay draws fresh noise (Philox offsets advance under replay)
    gd = _diffusion(1000)
    a = gd.p_sample_loop(m, x0.shape, noise=x0.clone(), clip_denoised=False, model_kwargs=kw)
    assert torch.isfinite(a).all() and rel(a, graph) > 1e-3


def _single_person_trainer(c, m):
    args = types.SimpleNamespace(device=torch.device(DEV), diffusion_steps=1000, is_train=True, lr=2e-4, batch_size=c["B"],
                                 num_epochs=1, log_every=50, save_latest=500, save_every_e=5, is_continue=False, model_dir="/tmp")
    return hig_amd.DDPMTrainer(args, m)


def _backlog():
    """About a quarter of a second of queued device work: what follows is enqueued onto a BUSY device."""
    a = torch.randn(8192, 8192, device=DEV)
    for _ in range(12):
        a = (a @ a).mul_(1e-4)
    return a


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_captured_single_person_step_does_not_depend_on_when_the_host_synchronises(storage):
    """The round-5 two-person regression (tests/test_gpu_interaction.py) as a check of EVERY captured training form (VERDICT r05
    item 8): the single-person step at BASELINE config 2's size, replayed (a) back to back, (b) onto an idle device after a
    torch.cuda.synchronize() -- bench.py's timed() pattern -- and (c) behind a backlog of unrelated work.  A graph node that only
    happens to be ordered by the host's enqueue timing shows as a different loss trajectory; the three must be bit-identical."""
    import time
    c = CONFIG2
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(c["B"], c["T"], c["F"], generator=g).to(DEV)
    nz = torch.randn(c["B"], c["T"], c["F"], generator=g).to(DEV)
    tt = torch.tensor(c["t"], device=DEV)
    ln = torch.tensor(c["lengths"], device=DEV)
    xp = torch.randn(c["B"], 4 * c["d"], generator=g).to(DEV)
    xo = torch.randn(c["B"], c["N"], c["Lt"], generator=g).to(DEV)
    runs = []
    for pattern in ("back to back", "idle", "backlog"):
        m = build(c, storage=storage).train()
        tr = _single_person_trainer(c, m)
        losses = []
        for k in range(6):
            if pattern == "backlog" and k >= 2:
                _backlog()
            tr.train_step_captured(x0, tt, ln, xp, xo, noise=nz)
            losses.append(tr.fused_state()["loss"].clone())
            if pattern == "idle" and k in (1, 3):
                torch.cuda.synchronize()
                time.sleep(0.05)
        runs.append([v.item() for v in losses])
        del tr, m
    assert runs[0] == runs[1] == runs[2], runs
    assert all(v == v and abs(v) < float("inf") for v in runs[0]), runs[0]
    assert runs[0][-1] < runs[0][0]          # and it trains: same batch six times


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_captured_sampling_loop_replayed_onto_an_idle_device_equals_back_to_back(mode):
    """The same check for the captured p_sample_loop (one hipGraph per step, replayed `steps` times): 50 steps at config 3's
    batch with the noise draws zeroed, (a) right after capture, (b) again behind a backlog of unrelated device work, (c) again
    onto an idle device after a synchronisation.  Bit-identical outputs."""
    import time
    c = dict(fill.CASES["width"], B=32)
    m = build(c, precision=mode).eval()
    with torch.no_grad():
        m.out.weight.mul_(0.05)
        m.out.bias.mul_(0.05)
    g = torch.Generator().manual_seed(12)
    kw = {"xf_proj": torch.randn(32, 4 * c["d"], generator=g).to(DEV),
          "xf_out": torch.randn(32, c["N"], c["Lt"], generator=g).to(DEV),
          "length": torch.tensor([196] * 20 + list(range(100, 196, 8)), device=DEV)}
    x0 = torch.randn(32, 196, c["F"], generator=g).to(DEV)
    gd = _diffusion(50)
    first = _zero_noise_loop(m, gd, True, x0, kw).clone()
    _backlog()
    busy = _zero_noise_loop(m, gd, True, x0, kw).clone()
    torch.cuda.synchronize()
    time.sleep(0.05)
    idle = _zero_noise_loop(m, gd, True, x0, kw).clone()
    assert torch.isfinite(first).all()
    assert torch.equal(first, busy) and torch.equal(first, idle)
