"""Whole-path parity on the MI355X: the HIP denoiser / diffusion / trainer against (a) golden
vectors produced by the reference itself (tests/golden, oracle/make_golden.py) and (b) the CPU
oracle on the same seeded inputs.  Tolerance: north_star's 1e-3 relative fp32 is the gate; the
asserts below are tighter where fp32 accumulation allows."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import hig_amd  # noqa: E402
from hig_amd.models import gaussian_diffusion as gdm  # noqa: E402
from oracle import denoiser_ref as R  # noqa: E402
from oracle import diffusion_ref as D  # noqa: E402
from oracle import fill  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def build(c, no_eff=False):
    m = hig_amd.MotionTransformer(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"],
                                  ff_size=c["ff"], num_layers=c["L"], num_heads=c["H"],
                                  text_latent_dim=c["Lt"], no_eff=no_eff)
    m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
    return m.to(DEV)


def case_inputs(c):
    inp = fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"])
    return inp, {k: v.to(DEV) for k, v in inp.items()}


def make_diffusion(n):
    return hig_amd.GaussianDiffusion(betas=gdm.get_named_beta_schedule("linear", n),
                                     model_mean_type=gdm.ModelMeanType.EPSILON,
                                     model_var_type=gdm.ModelVarType.FIXED_SMALL,
                                     loss_type=gdm.LossType.MSE)


@pytest.mark.parametrize("case", ["tiny", "config1", "width"])
def test_forward_matches_reference_golden(gold, case):
    g = gold("g2_denoiser_fwd.npz")
    c = fill.CASES[case]
    m = build(c).eval()
    _, gi = case_inputs(c)
    with torch.no_grad():
        out = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
    assert out.shape == (c["B"], c["T"], c["F"])
    assert rel(out, g[case + ".lin.out"]) < 2e-5


@pytest.mark.parametrize("precision,tol", [("bf16x3", 5e-5), ("bf16", 3e-2)])
def test_forward_reduced_product_modes(gold, precision, tol):
    """Split-bf16 products keep the fp32 parity gate (1e-3) with two orders of margin; plain bf16
    products are reported, not gated at 1e-3."""
    g = gold("g2_denoiser_fwd.npz")
    for case in ("config1", "width"):
        c = fill.CASES[case]
        m = build(c).eval()
        m.precision = precision
        _, gi = case_inputs(c)
        with torch.no_grad():
            out = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
        e = rel(out, g[case + ".lin.out"])
        assert e < tol, (case, e)
        if precision == "bf16":
            assert e > 1e-4  # really took the bf16 path


@pytest.mark.parametrize("case", ["tiny", "config1", "width"])
def test_no_eff_forward_matches_reference_golden(gold, case):
    """no_eff=True (full softmax attention, query-axis mask quirk).  Rows of full-length samples and
    the valid rows of padded samples are checked tightly; padded QUERY rows only loosely -- there the
    reference itself carries logits quantised to 2^-7 by its -1e5 offset (its own
    fp32-vs-fp64 error is 1.9e-4 on this variant)."""
    g = gold("g2_denoiser_fwd.npz")
    c = fill.CASES[case]
    m = build(c, no_eff=True).eval()
    _, gi = case_inputs(c)
    with torch.no_grad():
        out = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"]).cpu()
    ref = torch.tensor(g[case + ".full.out"])
    n = c["lengths"][1]
    assert rel(out[0], ref[0]) < 1e-3 and rel(out[1, :n], ref[1, :n]) < 1e-3      # the north-star gate
    assert rel(out, ref) < 1e-2
    assert rel(out[0], ref[0]) < 2e-4


@pytest.mark.parametrize("case", ["tiny", "config1"])
def test_no_eff_backward_matches_reference_golden(gold, case):
    g = gold("g3_denoiser_bwd.npz")
    c = fill.CASES[case]
    tag = case + ".full"
    m = build(c, no_eff=True).train()
    _, gi = case_inputs(c)
    x, xp, xo = (gi[k].clone().requires_grad_(True) for k in ("x", "xf_proj", "xf_out"))
    out = m(x, gi["t"], length=gi["length"], xf_proj=xp, xf_out=xo)
    r = (fill.tensor_for("loss.r." + case, out.shape) * 10.0).to(DEV)
    (out * r).sum().backward()
    # gradients flow through the padded query rows too, whose quantised logits make both sides noisy
    assert rel(x.grad, g[tag + ".dx"]) < 2e-2
    assert rel(xp.grad, g[tag + ".dxf_proj"]) < 2e-2
    assert rel(xo.grad, g[tag + ".dxf_out"]) < 2e-2
    named = dict(m.named_parameters())
    core = fill.core_param_shapes(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"])
    tot = torch.sqrt(sum((named[k].grad.double() ** 2).sum() for k in core)).item()
    ref = float(g[tag + ".gnorm_core"])
    assert abs(tot - ref) / ref < 1e-2
    # a full-length batch has no quantised rows: tight check against the CPU oracle
    p = {k: v.clone().requires_grad_(True) for k, v in
         fill.core_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"]).items()}
    inp, _ = case_inputs(c)
    full_len = torch.full((c["B"],), c["T"], dtype=torch.int64)
    xr = inp["x"].clone().requires_grad_(True)
    o_ref = R.denoiser_forward(p, xr, inp["t"], full_len, inp["xf_proj"], inp["xf_out"], c["H"], c["L"], no_eff=True)
    (o_ref * r.cpu()).sum().backward()
    m.zero_grad()
    x2 = gi["x"].clone().requires_grad_(True)
    o2 = m(x2, gi["t"], length=full_len.to(DEV), xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
    (o2 * r).sum().backward()
    assert rel(o2, o_ref) < 2e-5 and rel(x2.grad, xr.grad) < 1e-4
    for k in ("out.weight", "temporal_decoder_blocks.0.sa_block.query.weight",
              "temporal_decoder_blocks.0.sa_block.key.weight", "temporal_decoder_blocks.0.ca_block.value.weight",
              "temporal_decoder_blocks.0.ca_block.value.bias"):  # key.bias grads are identically 0
        assert rel(named[k].grad, p[k].grad) < 1e-4, k


def test_no_eff_head_dim_128_long_sequence_against_oracle():
    """no_eff attention on the matrix cores: head dim 128, T = 300 (transformer.py:196-285 has no head-dim
    limit).  Forward on the valid rows of a ragged batch, forward + backward tight on a full-length batch."""
    c = dict(B=2, T=300, F=20, d=256, H=2, L=2, ff=256, N=77, Lt=64, num_frames=300)
    m = build(c, no_eff=True).train()
    p = {k: v.clone().requires_grad_(True) for k, v in
         fill.core_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"]).items()}
    for lens, tight in (((300, 123), False), ((300, 300), True)):
        inp = fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], lens, (17, 803))
        gi = {k: v.to(DEV) for k, v in inp.items()}
        m.zero_grad()
        x = gi["x"].clone().requires_grad_(True)
        out = m(x, gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
        xr = inp["x"].clone().requires_grad_(True)
        ref = R.denoiser_forward(p, xr, inp["t"], inp["length"], inp["xf_proj"], inp["xf_out"], c["H"], c["L"], no_eff=True)
        for b, n in enumerate(lens):
            assert rel(out[b, :n], ref[b, :n]) < (2e-5 if tight else 1e-3), (lens, b)
        if not tight:
            continue
        r = (fill.tensor_for("hd128.r", out.shape) * 10.0)
        for v in p.values():
            v.grad = None
        (out * r.to(DEV)).sum().backward()
        (ref * r).sum().backward()
        assert rel(x.grad, xr.grad) < 1e-4
        named = dict(m.named_parameters())
        for k in ("out.weight", "temporal_decoder_blocks.0.sa_block.query.weight", "temporal_decoder_blocks.1.sa_block.key.weight",
                  "temporal_decoder_blocks.0.sa_block.value.weight", "temporal_decoder_blocks.1.ca_block.query.weight",
                  "temporal_decoder_blocks.0.ca_block.value.weight", "temporal_decoder_blocks.0.ca_block.key.weight"):
            assert rel(named[k].grad, p[k].grad) < 1e-4, k


def test_forward_intermediates_layer0(gold):
    """Layer-0 block outputs through the per-kernel ABI against the reference's hooks (tiny)."""
    g = gold("g2_denoiser_fwd.npz")
    c = fill.CASES["tiny"]
    p = fill.core_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"])
    inp, gi = case_inputs(c)
    with torch.no_grad():
        _, inter = R.denoiser_forward(p, inp["x"], inp["t"], inp["length"], inp["xf_proj"], inp["xf_out"],
                                      c["H"], c["L"], return_intermediates=True)
    for k in ("sa0", "ca0", "ffn0"):
        assert rel(inter[k], g["tiny.lin." + k]) < 2e-6  # oracle == reference (sanity, CPU)
    # 1-layer model on the GPU == reference ffn0 after its `out` is replaced by identity-free check:
    c1 = dict(c, L=1)
    m = build(c1).eval()
    with torch.no_grad():
        out = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
        ref = torch.nn.functional.linear(torch.tensor(g["tiny.lin.ffn0"]), p["out.weight"], p["out.bias"])
    assert rel(out, ref) < 2e-5


@pytest.mark.parametrize("case", ["tiny", "config1"])
def test_backward_matches_reference_golden(gold, case):
    g = gold("g3_denoiser_bwd.npz")
    c = fill.CASES[case]
    tag = case + ".lin"
    m = build(c).train()
    _, gi = case_inputs(c)
    x, xp, xo = (gi[k].clone().requires_grad_(True) for k in ("x", "xf_proj", "xf_out"))
    out = m(x, gi["t"], length=gi["length"], xf_proj=xp, xf_out=xo)
    r = (fill.tensor_for("loss.r." + case, out.shape) * 10.0).to(DEV)
    (out * r).sum().backward()
    assert rel(x.grad, g[tag + ".dx"]) < 5e-5
    assert rel(xp.grad, g[tag + ".dxf_proj"]) < 5e-5
    assert rel(xo.grad, g[tag + ".dxf_out"]) < 5e-5
    named = dict(m.named_parameters())
    checked = 0
    for k in g.files:
        if not k.startswith(tag + ".g."):
            continue
        a = named[k[len(tag) + 3:]].grad.double().cpu()
        b = torch.as_tensor(g[k]).double()
        # key.bias gradients are mathematically zero: relative + absolute floor
        assert (a - b).norm() <= 5e-5 * b.norm() + 2e-5 * b.numel() ** 0.5, k
        checked += 1
    assert checked >= 20
    core = fill.core_param_shapes(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"])
    tot = torch.sqrt(sum((named[k].grad.double() ** 2).sum() for k in core)).item()
    ref = float(g[tag + ".gnorm_core"])
    assert abs(tot - ref) / ref < 5e-5


def test_backward_with_split_bf16_products(gold):
    """precision="bf16x3": forward, data-gradient and weight-gradient GEMMs all run the split-bf16
    MFMA path (transposed operands); gradients stay within 1e-3 of the reference's."""
    g = gold("g3_denoiser_bwd.npz")
    c = fill.CASES["config1"]
    tag = "config1.lin"
    m = build(c).train()
    m.precision = "bf16x3"
    _, gi = case_inputs(c)
    x, xp, xo = (gi[k].clone().requires_grad_(True) for k in ("x", "xf_proj", "xf_out"))
    out = m(x, gi["t"], length=gi["length"], xf_proj=xp, xf_out=xo)
    r = (fill.tensor_for("loss.r.config1", out.shape) * 10.0).to(DEV)
    (out * r).sum().backward()
    assert rel(x.grad, g[tag + ".dx"]) < 1e-3
    assert rel(xp.grad, g[tag + ".dxf_proj"]) < 1e-3 and rel(xo.grad, g[tag + ".dxf_out"]) < 1e-3
    named = dict(m.named_parameters())
    core = fill.core_param_shapes(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"])
    tot = torch.sqrt(sum((named[k].grad.double() ** 2).sum() for k in core)).item()
    ref = float(g[tag + ".gnorm_core"])
    assert abs(tot - ref) / ref < 1e-3
    for k in g.files:
        if k.startswith(tag + ".g.") and not k.endswith("key.bias"):
            assert rel(named[k[len(tag) + 3:]].grad, g[k]) < 1e-3, k


def test_config5_shape_head_dim_128_long_sequence():
    """BASELINE config 5 geometry at small batch/depth: T=300, d=1024, H=8 (hd=128), ff=1024.
    Exercises the hd=128 linear-attention templates, NIT=4 row kernels and E=4096 GEMMs; forward
    and backward against the CPU oracle."""
    c = dict(B=2, T=300, F=150, d=1024, H=8, L=2, ff=1024, N=77, Lt=256, num_frames=300,
             lengths=(300, 173), t=(12, 901))
    m = build(c).train()
    inp, gi = case_inputs(c)
    x = gi["x"].clone().requires_grad_(True)
    out = m(x, gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
    r = (fill.tensor_for("loss.r.c5", out.shape) * 10.0).to(DEV)
    (out * r).sum().backward()
    p = {k: v.requires_grad_(True) for k, v in
         fill.core_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"]).items()}
    xr = inp["x"].clone().requires_grad_(True)
    ref = R.denoiser_forward(p, xr, inp["t"], inp["length"], inp["xf_proj"], inp["xf_out"], c["H"], c["L"])
    (ref * r.cpu()).sum().backward()
    assert rel(out, ref) < 2e-5 and rel(x.grad, xr.grad) < 1e-4
    named = dict(m.named_parameters())
    for k in ("out.weight", "temporal_decoder_blocks.0.sa_block.query.weight",
              "temporal_decoder_blocks.0.sa_block.key.weight", "temporal_decoder_blocks.1.ca_block.value.weight",
              "temporal_decoder_blocks.1.ffn.linear1.weight", "temporal_decoder_blocks.0.sa_block.proj_out.emb_layers.1.weight",
              "sequence_embedding"):
        assert rel(named[k].grad, p[k].grad) < 2e-4, k
    m.precision = "bf16"
    with torch.no_grad():
        o16 = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
    assert 1e-4 < rel(o16, ref) < 3e-2


def test_properties_padding_permutation_zero_init():
    c = fill.CASES["config1"]
    m = build(c).eval()
    _, gi = case_inputs(c)
    with torch.no_grad():
        full = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
        n = c["lengths"][1]
        tr = m(gi["x"][1:2, :n].contiguous(), gi["t"][1:2], length=torch.tensor([n]),
               xf_proj=gi["xf_proj"][1:2], xf_out=gi["xf_out"][1:2].contiguous())
        assert rel(tr, full[1:2, :n]) < 1e-5          # padding invariance (linear attention)
        perm = torch.tensor([1, 0], device=DEV)
        pr = m(gi["x"][perm], gi["t"][perm], length=gi["length"][perm], xf_proj=gi["xf_proj"][perm],
               xf_out=gi["xf_out"][perm])
        assert rel(pr, full[perm]) < 1e-6              # batch-permutation equivariance
        fresh = hig_amd.MotionTransformer(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"],
                                          ff_size=c["ff"], num_layers=c["L"], num_heads=c["H"],
                                          text_latent_dim=c["Lt"]).to(DEV).eval()
        z = fresh(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
        assert torch.count_nonzero(z) == 0             # zero_module init => output == 0
        # generate_src_mask contract: CPU float (B, T)
        msk = m.generate_src_mask(c["T"], gi["length"])
        assert msk.device.type == "cpu" and msk.shape == (c["B"], c["T"]) and msk.sum().item() == sum(c["lengths"])


def test_full_size_forward_against_oracle():
    """BASELINE config 2 shape (B=64, T=196, d=512, L=8): GPU vs CPU oracle, 1e-3 gate."""
    c = dict(fill.CASES["width"], B=64, lengths=tuple([196] * 32 + list(range(40, 72))),
             t=tuple(int(v) for v in np.linspace(0, 999, 64)))
    m = build(c).eval()
    inp, gi = case_inputs(c)
    with torch.no_grad():
        out = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
        torch.set_num_threads(max(1, torch.get_num_threads()))
        p = fill.core_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"])
        ref = R.denoiser_forward(p, inp["x"], inp["t"], inp["length"], inp["xf_proj"], inp["xf_out"],
                                 c["H"], c["L"])
    e = rel(out, ref)
    mx = ((out.cpu() - ref).abs().max() / ref.abs().max()).item()
    assert e < 1e-3 and mx < 1e-3, (e, mx)
    assert e < 5e-5  # what fp32 MFMA accumulate actually delivers


class _NoiseFeed:
    def __init__(self, prefix, dev):
        self.prefix, self.i, self.dev = prefix, 0, dev

    def _next(self, shape):
        v = (fill.tensor_for("%s.%d" % (self.prefix, self.i), shape) * 10.0).to(self.dev)
        self.i += 1
        return v

    def randn(self, *shape, device=None, **_):
        return self._next(shape)

    def randn_like(self, x, **_):
        return self._next(x.shape)


def _patch(feed):
    proxy = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
    proxy.randn, proxy.randn_like = feed.randn, feed.randn_like
    old = gdm.th
    gdm.th = proxy
    return lambda: setattr(gdm, "th", old)


def test_sampling_loop_matches_reference_golden(gold):
    g = gold("g5_loop.npz")
    c = fill.CASES["tiny"]
    m = build(c).eval()
    _, gi = case_inputs(c)
    gd = make_diffusion(50)
    gd.use_hip_graph = False   # injected noise sequence: the eager loop, step for step
    undo = _patch(_NoiseFeed("g5.z", DEV))
    try:
        final = gd.p_sample_loop(m, (c["B"], c["T"], c["F"]), clip_denoised=False,
                                 model_kwargs={"xf_proj": gi["xf_proj"], "xf_out": gi["xf_out"],
                                               "length": gi["length"]})
    finally:
        undo()
    assert rel(final, g["final"]) < 2e-4  # 50 chained steps


def test_graph_captured_loop_equals_eager_loop():
    c = fill.CASES["tiny"]
    m = build(c).eval()
    _, gi = case_inputs(c)
    kw = {"xf_proj": gi["xf_proj"], "xf_out": gi["xf_out"], "length": gi["length"]}
    x0 = (fill.tensor_for("graph.x0", (c["B"], c["T"], c["F"])) * 10.0).to(DEV)
    outs = []
    for use_graph in (False, True):
        gd = make_diffusion(50)
        gd.use_hip_graph = use_graph
        gd._debug_zero_noise = True
        if not use_graph:
            proxy = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
            proxy.randn_like = lambda x, **_: torch.zeros_like(x)
            old, gdm.th = gdm.th, proxy
        try:
            outs.append(gd.p_sample_loop(m, x0.shape, noise=x0.clone(), clip_denoised=False, model_kwargs=kw))
        finally:
            if not use_graph:
                gdm.th = old
    assert torch.isfinite(outs[1]).all()
    assert rel(outs[1], outs[0]) < 1e-6
    # and with real noise the graph path draws fresh noise every replay
    gd = make_diffusion(50)
    a = gd.p_sample_loop(m, x0.shape, noise=x0.clone(), clip_denoised=False, model_kwargs=kw)
    assert torch.isfinite(a).all() and rel(a, outs[0]) > 1e-3


def _trainer(c, m):
    args = types.SimpleNamespace(device=torch.device(DEV), diffusion_steps=1000, is_train=True, lr=2e-4,
                                 batch_size=c["B"], num_epochs=1, log_every=50, save_latest=500,
                                 save_every_e=5, is_continue=False, model_dir="/tmp")
    return hig_amd.DDPMTrainer(args, m)


def test_trainer_step_matches_reference_golden(gold):
    """DDPMTrainer.forward + update (reference sequence, stub CLIP + torch text head) == G6."""
    import hig_amd.trainers.ddpm_trainer as tr
    g = gold("g6_trainer.npz")
    c = fill.CASES["config1"]
    m = build(c).train()
    trainer = _trainer(c, m)
    trainer.opt_encoder = torch.optim.Adam(m.parameters(), lr=2e-4)
    motions = fill.tensor_for("g6.motions", (c["B"], c["T"], c["F"])) * 10
    captions = ["a person shakes hands with another person", "two people hug"]
    t_fixed = torch.tensor(c["t"])
    trainer.sampler.sample = lambda bs, dev: (t_fixed.to(dev), torch.ones(bs))
    captured = {}
    real_clip = tr.clip_grad_norm_

    def spy(params, max_norm):
        captured["gnorm"] = float(real_clip(params, max_norm))
        return captured["gnorm"]

    tr.clip_grad_norm_ = spy
    undo = _patch(_NoiseFeed("g6.noise", DEV))
    try:
        trainer.forward((captions, motions, torch.tensor(c["lengths"])))
        logs = trainer.update()
    finally:
        undo()
        tr.clip_grad_norm_ = real_clip
    assert rel(trainer.fake_noise, g["fake_noise"]) < 5e-5
    assert np.array_equal(trainer.src_mask.cpu().numpy(), g["src_mask"])
    assert abs(logs["loss_mot_rec"] - float(g["loss_mot_rec"])) < 1e-4 * float(g["loss_mot_rec"])
    assert abs(captured["gnorm"] - float(g["gnorm"])) < 1e-4 * float(g["gnorm"])
    sd = m.state_dict()
    for k in g.files:
        if k.startswith("p."):
            # Adam's first step moves every weight by ~lr regardless of gradient scale
            assert (sd[k[2:]].cpu() - torch.tensor(g[k])).abs().max().item() < 2e-5, k


def test_fused_train_step_equals_reference_sequence():
    """train_step_fused (HIP masked-MSE + flat clip+Adam) == forward()/update() on core params."""
    c = fill.CASES["config1"]
    _, gi = case_inputs(c)
    x0 = (fill.tensor_for("fused.x0", (c["B"], c["T"], c["F"])) * 10).to(DEV)
    noise = (fill.tensor_for("fused.noise", x0.shape) * 10).to(DEV)
    # reference sequence through autograd, core parameters only
    m1 = build(c).train()
    gd = make_diffusion(1000)
    opt = torch.optim.Adam(m1.core_parameters(), lr=2e-4)
    out = gd.training_losses(m1, x0, gi["t"], model_kwargs={"xf_proj": gi["xf_proj"], "xf_out": gi["xf_out"],
                                                            "length": gi["length"]}, noise=noise)
    mask = m1.generate_src_mask(c["T"], gi["length"]).to(DEV)
    loss = (((out["pred"] - out["target"]) ** 2).mean(-1) * mask).sum() / mask.sum()
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(m1.core_parameters(), 0.5)
    opt.step()
    # fused
    m2 = build(c).train()
    tr2 = _trainer(c, m2)
    l2 = tr2.train_step_fused(x0, gi["t"], gi["length"], gi["xf_proj"], gi["xf_out"], noise=noise)
    st = tr2.fused_state()
    assert abs(l2.item() - loss.item()) < 1e-5 * abs(loss.item())
    assert abs(st["gnorm"].item() - gn.item()) < 1e-4 * gn.item()
    assert st["step"].item() == 1
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert (a - b).abs().max().item() < 2e-6, k


def test_captured_train_step_equals_eager_fused_step():
    """hipGraph-captured training step == the eager fused step, over 3 consecutive steps with
    fresh inputs each step (graph replay must pick up the staged inputs and the step counter)."""
    c = fill.CASES["config1"]
    _, gi = case_inputs(c)
    m1, m2 = build(c).train(), build(c).train()
    t1, t2 = _trainer(c, m1), _trainer(c, m2)
    for it in range(3):
        x0 = (fill.tensor_for("cap.x0.%d" % it, (c["B"], c["T"], c["F"])) * 10).to(DEV)
        noise = (fill.tensor_for("cap.noise.%d" % it, x0.shape) * 10).to(DEV)
        tt = torch.tensor([(37 * it + 5) % 1000, (411 * it + 900) % 1000], device=DEV)
        l1 = t1.train_step_fused(x0, tt, gi["length"], gi["xf_proj"], gi["xf_out"], noise=noise).clone()
        l2 = t2.train_step_captured(x0, tt, gi["length"], gi["xf_proj"], gi["xf_out"], noise=noise).clone()
        assert abs(l1.item() - l2.item()) <= 1e-6 * abs(l1.item())
    assert t2.fused_state()["step"].item() == 3
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k                      # same kernels, same order: bit-identical


def test_trainer_train_loop_generate_and_resume(tmp_path):
    """DDPMTrainer.train (epoch loop, sharded loader, checkpoints), .generate (hipGraph sampling through
    encode_text) and --is_continue resume: the reference's tool-level call sequence
    (tools/train.py:86-88, tools/visualization.py:163-186) on a synthetic dataset."""
    c = fill.CASES["tiny"]
    torch.manual_seed(0)

    class Toy(torch.utils.data.Dataset):
        caps = ["a person waves", "two people shake hands", "someone walks forward", "a person jumps"]

        def __len__(self):
            return 8

        def __getitem__(self, i):
            g = torch.Generator().manual_seed(i)
            return self.caps[i % 4], torch.randn(c["T"], c["F"], generator=g), c["T"] - (i % 3)

    m = build(c).train()
    tr = _trainer(c, m)
    tr.opt.model_dir = str(tmp_path)
    tr.opt.num_epochs, tr.opt.batch_size, tr.opt.log_every, tr.opt.num_workers = 2, 4, 2, 0
    before = m.out.weight.detach().clone()
    tr.train(Toy(), rank=0, world_size=1)
    assert not torch.equal(before, m.out.weight) and torch.isfinite(m.out.weight).all()
    ck = torch.load(str(tmp_path / "latest.tar"))
    assert ck["ep"] == 1 and ck["total_it"] == 4 and (tmp_path / "ckpt_e000.tar").exists()
    outs = tr.generate(["a person waves", "two people hug", "a person jumps"], torch.tensor([16, 12, 9]),
                       c["F"], batch_size=2)
    assert len(outs) == 3 and all(o.shape == (16, c["F"]) or o.shape[1] == c["F"] for o in outs)
    assert all(torch.isfinite(o).all() for o in outs)
    # resume
    m2 = build(c).train()
    tr2 = _trainer(c, m2)
    tr2.opt.model_dir, tr2.opt.is_continue = str(tmp_path), True
    tr2.opt.num_epochs, tr2.opt.batch_size, tr2.opt.log_every, tr2.opt.num_workers = 2, 4, 2, 0
    tr2.train(Toy(), rank=0, world_size=1)     # epochs 1..1: one more epoch from the checkpoint
    assert torch.load(str(tmp_path / "latest.tar"))["total_it"] == 6


def test_checkpoint_roundtrip_and_reference_keys(gold, tmp_path):
    g = gold("g7_state_dict_keys.npz")
    c = fill.CASES["tiny"]
    m = build(c)
    sd = m.state_dict()
    assert set(sd.keys()) == set(g.files)               # identical key set to the reference module
    for k in g.files:
        assert tuple(sd[k].shape) == tuple(g[k]), k
    trainer = _trainer(c, m)
    trainer.opt_encoder = torch.optim.Adam(m.parameters(), lr=2e-4)
    f = str(tmp_path / "latest.tar")
    trainer.save(f, 3, 17)
    ck = torch.load(f)
    assert set(ck.keys()) == {"opt_encoder", "ep", "total_it", "encoder"}
    m2 = build(dict(c))
    with torch.no_grad():
        for p in m2.parameters():
            p.add_(1.0)
    t2 = _trainer(c, m2)
    t2.opt_encoder = torch.optim.Adam(m2.parameters(), lr=2e-4)
    assert t2.load(f) == (3, 17)
    _, gi = case_inputs(c)
    with torch.no_grad():
        a = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
        b = m2(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
    assert torch.equal(a, b)


@pytest.mark.parametrize("no_eff", [False, True])
def test_edge_cases_empty_sample_single_frame_batch_of_one(no_eff):
    """Ragged extremes against the oracle, forward and backward: a sample with length 0 (every key masked),
    length 1, T not a multiple of anything, a single frame (T = 1), batch of one, lengths given past T."""
    base = dict(F=12, d=64, H=8, L=2, ff=128, N=77, Lt=32, num_frames=70)
    for (B, T, lens) in ((3, 67, (0, 1, 67)), (1, 1, (1,)), (2, 5, (9, 2))):
        c = dict(base, B=B, T=T)
        m = build(c, no_eff=no_eff).train()
        inp = fill.inputs(B, T, c["F"], c["d"], c["N"], c["Lt"], lens, (0, 999, 431)[:B])
        gi = {k: v.to(DEV) for k, v in inp.items()}
        x = gi["x"].clone().requires_grad_(True)
        out = m(x, gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
        r = fill.tensor_for("edge.r.%d.%d" % (B, T), out.shape) * 10.0
        (out * r.to(DEV)).sum().backward()
        p = {k: v.double().requires_grad_(True) for k, v in
             fill.core_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"]).items()}
        xr = inp["x"].double().requires_grad_(True)
        ref = R.denoiser_forward(p, xr, inp["t"], inp["length"], inp["xf_proj"].double(), inp["xf_out"].double(),
                                 c["H"], c["L"], no_eff=no_eff)
        assert torch.isfinite(out).all()
        if no_eff:
            # padded QUERY rows of the full-attention variant carry logits quantised by the reference's -1e5 offset
            # (App. B-3): gate the valid rows, and the rest loosely
            for b, n in enumerate(lens):
                n = min(n, T)
                if n:
                    assert rel(out[b, :n], ref[b, :n]) < 1e-3, (B, T, b)
            continue
        assert rel(out, ref) < 1e-5, (B, T, rel(out, ref))
        (ref * r.double()).sum().backward()
        assert rel(x.grad, xr.grad) < 1e-4
        named = dict(m.named_parameters())
        for k in ("temporal_decoder_blocks.0.sa_block.value.weight", "temporal_decoder_blocks.1.ffn.linear1.bias",
                  "sequence_embedding", "out.weight"):
            assert (named[k].grad.cpu().double() - p[k].grad).norm() <= 1e-4 * p[k].grad.norm() + 1e-9, (k, B, T)


@pytest.mark.parametrize("two_person", [False, True])
def test_workspaces_are_not_overrun(two_person):
    """Every scratch buffer the library asks for (sizes from hig_*_bytes) is handed out inside a larger
    sentinel-filled allocation; after forward + backward (+ text head) the guard bands must be intact."""
    guards = []

    def guarded_take(kind, nbytes, device):
        pad = 4096
        big = torch.full((int(nbytes) + 2 * pad,), 0xA5, dtype=torch.uint8, device=device)
        guards.append((kind, big, pad, int(nbytes)))
        return big[pad:pad + max(int(nbytes), 16)]

    if two_person:
        c = fill.ICASES["config1x2"]
        m = hig_amd.MotionInteractionTransformer(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"],
                                                 ff_size=c["ff"], num_layers=c["L"], num_heads=c["H"],
                                                 text_latent_dim=c["Lt"])
        m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
        m = m.to(DEV).train()
        inp = fill.inputs(2 * c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"] * 2, c["t"] * 2)
    else:
        c = fill.CASES["config1"]
        m = build(c).train()
        inp = fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"])
    gi = {k: v.to(DEV) for k, v in inp.items()}
    m._pool.take = guarded_take
    m._pool.give = lambda *a, **k: None
    caps = ["two people shake hands", "one pushes the other", "they hug", "a person waves"][:gi["x"].shape[0]]
    xf_proj, xf_out = m.encode_text(caps, DEV)                      # text head forward (training workspace)
    x = gi["x"].clone().requires_grad_(True)
    out = m(x, gi["t"], length=gi["length"], xf_proj=xf_proj, xf_out=xf_out)
    out.square().mean().backward()                                   # denoiser backward + text head backward
    with torch.no_grad():
        m(gi["x"], gi["t"], length=gi["length"], xf_proj=xf_proj.detach(), xf_out=xf_out.detach())   # inference workspaces
    torch.cuda.synchronize()
    kinds = {k for k, *_ in guards}
    assert {"fwd_t", "bwd", "textctx_t", "txt_t", "txt_bwd", "fwd_i"} <= kinds, kinds
    for kind, big, pad, n in guards:
        assert bool((big[:pad] == 0xA5).all()) and bool((big[pad + max(n, 16):] == 0xA5).all()), kind
    assert torch.isfinite(x.grad).all()


@pytest.mark.parametrize("case,B", [("config1", 2), ("width", 2), ("width", 30)])
def test_per_call_forward_with_the_batched_text_side_against_the_oracle_and_the_cached_form(case, B):
    """The reference computes the cross-attention text side on every call (transformer.py:144-150).  The fp32 inference forward
    does so in ONE key/value GEMM over the stacked, text_norm-folded weights of all layers and ONE context build
    (hig_denoiser_fwd_x with the derived table of models/transformer.py:_derived32): against the CPU oracle at the fp32 gate,
    against the cached form (per-layer GEMMs with a LayerNorm prologue: hig_text_context) to rounding, bitwise repeatable.
    B x 77 text rows: 154 (tiled GEMM), 2310 (the K = 256 instance of the weight-stationary kernel)."""
    c = dict(fill.CASES[case])
    if B != c["B"]:
        c.update(B=B, lengths=tuple(c["T"] - (7 * i) % (c["T"] - 1) for i in range(B)), t=tuple((37 * i) % 1000 for i in range(B)))
    m = build(c).eval()
    inp = fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"])
    gi = {k: v.to(DEV) for k, v in inp.items()}

    def fwd():
        with torch.no_grad():
            return m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])

    cached = fwd()
    m.cache_text_context = False
    per_call = fwd()
    torch.cuda.synchronize()
    assert m._derived32(m.flat_params())[6 * c["L"]] is not None          # the stacked text weights exist: the batched form ran
    assert torch.isfinite(per_call).all()
    assert rel(per_call, cached) < 2e-6
    for _ in range(3):
        assert torch.equal(fwd(), per_call)
    n = min(B, 2)
    p = fill.core_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"])
    with torch.no_grad():
        ref = R.denoiser_forward(p, inp["x"][:n], inp["t"][:n], inp["length"][:n], inp["xf_proj"][:n], inp["xf_out"][:n], c["H"], c["L"])
    assert rel(per_call[:n], ref) < 2e-5
