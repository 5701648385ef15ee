"""State handling of the fused training step on the MI355X: optimizer state in checkpoints (resume == uninterrupted
run, reference checkpoint layout ddpm_trainer.py:200-218), the text-context cache under in-place parameter updates,
captured graphs after the model is re-homed, and the configurations the fused step must refuse."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

import hig_amd  # noqa: E402
from oracle import fill  # noqa: E402

DEV = "cuda"
C1 = fill.CASES["config1"]


def build(c=C1):
    m = hig_amd.MotionTransformer(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"], ff_size=c["ff"],
                                  num_layers=c["L"], num_heads=c["H"], text_latent_dim=c["Lt"])
    m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
    return m.to(DEV)


def trainer(m, **opt):
    args = types.SimpleNamespace(device=torch.device(DEV), diffusion_steps=1000, is_train=True, lr=2e-4,
                                 batch_size=C1["B"], num_epochs=1, log_every=50, save_latest=500, save_every_e=5,
                                 is_continue=False, model_dir="/tmp", **opt)
    return hig_amd.DDPMTrainer(args, m)


def step_inputs(i):
    c = C1
    gi = {k: v.to(DEV) for k, v in fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"]).items()}
    x0 = (fill.tensor_for("state.x0.%d" % i, (c["B"], c["T"], c["F"])) * 10).to(DEV)
    nz = (fill.tensor_for("state.nz.%d" % i, x0.shape) * 10).to(DEV)
    t = torch.tensor([(91 * i + 7) % 1000, (613 * i + 400) % 1000], device=DEV)
    return x0, t, gi["length"], gi["xf_proj"], gi["xf_out"], nz


def run_steps(tr, steps, captured=False):
    for i in steps:
        x0, t, ln, xp, xo, nz = step_inputs(i)
        (tr.train_step_captured if captured else tr.train_step_fused)(x0, t, ln, xp, xo, noise=nz)


@pytest.mark.parametrize("captured", [False, True])
def test_fused_training_resumes_bit_exactly_from_a_checkpoint(tmp_path, captured):
    """train 3 fused steps, save, train 2 more  ==  load the checkpoint into a fresh trainer, train the same 2."""
    f = str(tmp_path / "latest.tar")
    m1 = build().train()
    t1 = trainer(m1, fused_step=True)
    run_steps(t1, range(3), captured)
    t1.save(f, 0, 3)
    run_steps(t1, range(3, 5), captured)
    ck = torch.load(f)
    assert set(ck) == {"opt_encoder", "ep", "total_it", "encoder"}
    st = ck["opt_encoder"]["state"]
    assert len(st) == len(m1.core_parameters()) and all(float(e["step"]) == 3.0 for e in st.values())
    assert all(e["exp_avg"].abs().sum() > 0 for e in list(st.values())[:5])       # real moments, not an empty Adam
    # the same file is a valid torch-Adam checkpoint (the reference's own trainer would load it)
    opt = torch.optim.Adam(build().parameters(), lr=2e-4)
    opt.load_state_dict(ck["opt_encoder"])

    m2 = build().train()
    with torch.no_grad():
        for p in m2.core_parameters():
            p.add_(0.5)                                   # overwritten by load()
    t2 = trainer(m2, fused_step=True)
    assert t2.load(f) == (0, 3)
    assert t2.fused_state()["step"].item() == 3
    run_steps(t2, range(3, 5), captured)
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert torch.equal(t1.fused_state()["m"], t2.fused_state()["m"]) and torch.equal(t1.fused_state()["v"], t2.fused_state()["v"])


def test_checkpoint_written_by_torch_adam_resumes_under_the_fused_step(tmp_path):
    """forward()/update() (torch Adam) for 2 steps -> save -> resume with opt.fused_step: the fused Adam continues from
    the torch moments (same update rule), so one more step matches torch Adam's third step."""
    f = str(tmp_path / "latest.tar")
    m1 = build().train()
    t1 = trainer(m1)
    t1.opt_encoder = torch.optim.Adam(m1.core_parameters(), lr=2e-4)

    def torch_step(tr, m, i):
        x0, t, ln, xp, xo, nz = step_inputs(i)
        out = tr.diffusion.training_losses(m, x0, t, model_kwargs={"xf_proj": xp, "xf_out": xo, "length": ln}, noise=nz)
        tr.real_noise, tr.fake_noise = out["target"], out["pred"]
        tr.src_mask = m.generate_src_mask(C1["T"], ln).to(DEV)
        tr.zero_grad([tr.opt_encoder])
        tr.backward_G()
        tr.loss_mot_rec.backward()
        torch.nn.utils.clip_grad_norm_(m.core_parameters(), 0.5)
        tr.step([tr.opt_encoder])

    for i in range(2):
        torch_step(t1, m1, i)
    # save() indexes optimizer state by position in encoder.parameters(): rebuild the optimizer over ALL of them
    full = torch.optim.Adam(m1.parameters(), lr=2e-4)
    idx = {id(p): i for i, p in enumerate(m1.parameters())}
    sd = full.state_dict()
    core_sd = t1.opt_encoder.state_dict()
    for j, p in enumerate(m1.core_parameters()):
        sd["state"][idx[id(p)]] = core_sd["state"][j]
    full.load_state_dict(sd)
    t1.opt_encoder = full
    t1.save(f, 0, 2)
    torch_step(t1, m1, 2)

    m2 = build().train()
    t2 = trainer(m2, fused_step=True)
    t2.load(f)
    run_steps(t2, [2])
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        if not k.startswith(("clip.", "text")):
            assert (a - b).abs().max().item() < 3e-6, k


def test_text_context_cache_follows_inplace_parameter_updates():
    """The inference text-context cache is keyed on the parameters it was built from: an in-place update through the
    nn.Parameters (torch optimizers, load_state_dict, p.copy_) must invalidate it although the flat buffer's own
    version counter does not move."""
    c = C1
    m = build().eval()
    gi = {k: v.to(DEV) for k, v in fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"]).items()}

    def fwd(model):
        with torch.no_grad():
            return model(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])

    a = fwd(m)
    assert m._textctx_cache is not None and torch.equal(fwd(m), a)       # second call served from the cache
    with torch.no_grad():
        m.temporal_decoder_blocks[1].ca_block.key.weight.mul_(1.5)
    b = fwd(m)
    assert not torch.equal(a, b)
    fresh = build().eval()
    with torch.no_grad():
        fresh.temporal_decoder_blocks[1].ca_block.key.weight.mul_(1.5)
    assert torch.equal(b, fwd(fresh))
    # (through autograd the forward keeps its LayerNorm kernels -- the backward needs their outputs and statistics -- while
    # the inference forward folds them into the projections: same function, rounding-level difference)
    tr_out = fresh(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"]).detach()
    assert ((tr_out - b).norm() / b.norm()).item() < 2e-6
    # load_state_dict and a torch optimizer step invalidate it too
    m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
    assert torch.equal(fwd(m), a)
    opt = torch.optim.SGD([m.temporal_decoder_blocks[0].ca_block.text_norm.weight], lr=0.1)
    m.temporal_decoder_blocks[0].ca_block.text_norm.weight.grad = torch.ones_like(m.temporal_decoder_blocks[0].ca_block.text_norm.weight)
    opt.step()
    assert not torch.equal(fwd(m), a)
    # cache switched off: every forward recomputes the text side inside the call (hig_denoiser_fwd_x; since round 6 in its batched
    # form -- one key/value GEMM over the stacked text_norm-folded weights, another rounding order of the same arithmetic): the
    # cached context's numbers to rounding, repeatably bit for bit, also right after the inputs change
    def close(u, v):
        return ((u - v).norm() / v.norm()).item() < 2e-6

    m.cache_text_context = True
    c0 = fwd(m)
    m.cache_text_context = False
    c1 = fwd(m)
    assert close(c1, c0) and torch.equal(c1, fwd(m))
    xo2 = gi["xf_out"] * 1.25
    with torch.no_grad():
        d1 = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=xo2)
        m.cache_text_context = True
        d0 = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=xo2)
    assert close(d1, d0) and not torch.equal(d1, c1)        # (text_norm makes a rescaled xf_out an eps-level change: not equal, but close)
    for _ in range(20):          # (the per-layer form's text-stream events, or the batched form's staging, are reused call after call)
        m.cache_text_context = False
        assert torch.equal(fwd(m), c1)


def test_captured_step_survives_rehoming_of_the_model():
    """`.to()` / `.float()` drop the flat parameter buffer; graphs captured against the old one must not be replayed
    (they would train orphaned memory): the captured step re-captures and keeps training the live parameters."""
    m1, m2 = build().train(), build().train()
    t1, t2 = trainer(m1), trainer(m2)
    run_steps(t1, [0], captured=True)
    run_steps(t2, [0], captured=False)
    t1.to(torch.device(DEV))          # what train() does first; re-homes the parameters into a NEW flat buffer
    m1.float()
    run_steps(t1, [1, 2], captured=True)
    run_steps(t2, [1, 2], captured=False)
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert t1.fused_state()["step"].item() == 3
    # the LR is a device scalar: a schedule does not grow the graph cache
    n_graphs = len(t1.fused_state()["graphs"])
    for i, lr in enumerate((1e-4, 5e-5, 2.5e-5)):
        x0, t, ln, xp, xo, nz = step_inputs(10 + i)
        t1.train_step_captured(x0, t, ln, xp, xo, noise=nz, lr=lr)
        t2.train_step_fused(x0, t, ln, xp, xo, noise=nz, lr=lr)
    assert len(t1.fused_state()["graphs"]) == n_graphs
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_fused_step_refuses_a_trainable_clip_tower():
    m = build().train()
    tr = trainer(m)
    caps = ["a person waves", "two people hug"]
    clip_out, eot = tr.clip_inputs(caps)
    x0, t, ln, _, _, nz = step_inputs(0)
    tr.train_step_fused(x0, t, ln, noise=nz, clip_out=clip_out, eot=eot)          # frozen CLIP: fine
    for p in m.clip.parameters():
        p.requires_grad_(True)
    with pytest.raises(NotImplementedError, match="CLIP"):
        tr.train_step_fused(x0, t, ln, noise=nz, clip_out=clip_out, eot=eot)


def test_shutdown_releases_and_recreates_the_side_stream():
    from hig_amd import _lib
    m = build().train()
    tr = trainer(m)
    run_steps(tr, [0])
    torch.cuda.synchronize()
    assert _lib.lib().hig_shutdown() == 0
    before = m.out.weight.detach().clone()
    run_steps(tr, [1])                 # the backward re-creates its second stream on demand
    torch.cuda.synchronize()
    assert not torch.equal(before, m.out.weight) and torch.isfinite(m.out.weight).all()
    assert _lib.lib().hig_shutdown() == 0


HOOK_CASES = {
    # name: (oracle case, two-person kind (0 single / 1 interaction block / 2 no_cross_attn), storage)
    "single-f32": ("width", 0, "f32"), "single-bf16": ("width", 0, "bf16"),
    "pair-f32": ("config1", 1, "f32"), "pair-bf16": ("width", 1, "bf16"), "pair-nocross-bf16": ("width", 2, "bf16"),
}


@pytest.mark.parametrize("name", list(HOOK_CASES))
def test_backward_with_layer_hook_equals_plain_backward(name):
    """hig_denoiser_bwd_hooked / hig_denoiser_bwd_bf16_hooked: the hook fires once per decoder layer, last layer first,
    with that layer's gradients final on the comm stream; the whole gradient buffer equals the plain backward's (the
    stylization emb_layers gradient is computed per layer instead of once at the end: same products, same order).
    Both storage modes, single-person and two-person models (4 stylization blocks per layer with the interaction
    block)."""
    case, two, storage = HOOK_CASES[name]
    c = dict(fill.CASES[case])
    if two:
        c["L"] = min(c["L"], 3)
        m = hig_amd.MotionInteractionTransformer(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"], ff_size=c["ff"],
                                                 num_layers=c["L"], num_heads=c["H"], text_latent_dim=c["Lt"],
                                                 no_cross_attn=(two == 2), storage=storage)
        m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
        m = m.to(DEV).train()
        T = c["T"]
        inp = fill.inputs(2 * c["B"], T, c["F"], c["d"], c["N"], c["Lt"], tuple(min(x, T - 1) for x in c["lengths"]) * 2, c["t"] * 2)
    else:
        m = build(c).train()
        m.storage = storage
        inp = fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"])
    nsty = 4 if two == 1 else 3
    gi = {k: v.to(DEV) for k, v in inp.items()}
    dout = (fill.tensor_for("hook.dout", gi["x"].shape) * 10).to(DEV)
    fp = m.flat_params()

    def run(hook=None, comm=None):
        out, saved = m._launch_forward(gi["x"], gi["t"], gi["length"], gi["xf_proj"], gi["xf_out"], training=True)
        fp.ensure_grad()
        fp.grad.fill_(float("nan"))
        m._launch_backward(gi["x"], gi["t"], gi["length"], gi["xf_out"], saved, dout, want_dx=False, layer_hook=hook,
                           comm_stream=comm)
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(g).all()) for g in fp.grad_views)       # every parameter gradient was written
        return torch.nan_to_num(fp.grad[:fp.core_numel], nan=0.0)               # (alignment gaps between groups are not)

    plain = run()
    per_layer, tail = fp.layer_buckets(c["L"], nsty, c["d"], 4 * c["d"])
    covered = torch.zeros(fp.core_numel, dtype=torch.bool)
    for a, b in [r for lay in per_layer for r in lay] + list(tail):
        assert not covered[a:b].any()                   # the buckets partition the flat buffer
        covered[a:b] = True
    assert covered.all()
    comm = torch.cuda.Stream()
    seen, snaps = [], {}

    def hook(l):
        seen.append(l)
        with torch.cuda.stream(comm):                  # what the exchange would read: layer l's ranges, on the comm stream
            snaps[l] = [torch.nan_to_num(fp.grad[a:b], nan=0.0) for a, b in per_layer[l]]

    hooked = run(hook, comm)
    assert seen == list(range(c["L"] - 1, -1, -1))
    assert torch.equal(hooked, plain)
    for l, parts in snaps.items():
        for (a, b), snap in zip(per_layer[l], parts):
            assert torch.equal(snap, plain[a:b]), (l, a, b)          # already final when the hook's stream got there
    with pytest.raises(RuntimeError, match="boom"):                  # a failing hook surfaces as a Python exception
        run(lambda l: (_ for _ in ()).throw(RuntimeError("boom")), comm)
