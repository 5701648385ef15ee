"""Two-person path on the MI355X: MotionInteractionTransformer / DDPMMulTrainer over the
HIP kernels against goldens produced by the reference itself (g8, g9) and the CPU oracle."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import hig_amd  # noqa: E402
from oracle import fill  # noqa: E402
from oracle import interaction_ref as IR  # noqa: E402
from test_gpu_denoiser import _NoiseFeed, _patch, make_diffusion, rel  # noqa: E402

DEV = "cuda"


def build(c, **kw):
    m = hig_amd.MotionInteractionTransformer(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"],
                                             ff_size=c["ff"], num_layers=c["L"], num_heads=c["H"],
                                             text_latent_dim=c["Lt"], **kw)
    m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
    return m.to(DEV)


def case_inputs(c):
    inp = fill.inputs(2 * c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"] * 2, c["t"] * 2)
    return inp, {k: v.to(DEV) for k, v in inp.items()}


@pytest.mark.parametrize("case", ["tiny2", "config1x2"])
@pytest.mark.parametrize("nocross", [False, True])
def test_forward_matches_reference_golden(gold, case, nocross):
    g = gold("g8_interaction.npz")
    c = fill.ICASES[case]
    m = build(c, no_cross_attn=nocross).eval()
    _, gi = case_inputs(c)
    with torch.no_grad():
        out = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
    assert out.shape == (2 * c["B"], c["T"], c["F"])
    ref = torch.tensor(g[case + (".nocross" if nocross else "") + ".out"])
    assert rel(out, ref) < 2e-5
    assert rel(out[:, 0], ref[:, 0]) < 2e-5      # init-pose rows (out2 head)


def test_state_dict_contract(gold):
    g = gold("g8_interaction.npz")
    c = fill.ICASES["tiny2"]
    sd = build(c).state_dict()
    keys = [k[5:] for k in g.files if k.startswith("keys.")]
    assert list(sd) == keys
    assert all(tuple(sd[k].shape) == tuple(g["keys." + k]) for k in keys)


@pytest.mark.parametrize("case", ["tiny2", "config1x2"])
def test_backward_matches_reference_golden(gold, case):
    g = gold("g8_interaction.npz")
    c = fill.ICASES[case]
    m = build(c).train()
    _, gi = case_inputs(c)
    x, xp, xo = (gi[k].clone().requires_grad_(True) for k in ("x", "xf_proj", "xf_out"))
    out = m(x, gi["t"], length=gi["length"], xf_proj=xp, xf_out=xo)
    r = (fill.tensor_for("loss.r." + case, out.shape) * 10.0).to(DEV)
    (out * r).sum().backward()
    assert rel(x.grad, g[case + ".dx"]) < 1e-4
    assert rel(x.grad[:, 0, :4], g[case + ".dx"][:, 0, :4]) < 1e-4       # through joint_embed2
    assert x.grad[:, 0, 4:].abs().max().item() == 0.0                     # unused init-pose features
    assert rel(xp.grad, g[case + ".dxf_proj"]) < 1e-4
    assert rel(xo.grad, g[case + ".dxf_out"]) < 1e-4
    named = dict(m.named_parameters())
    gnorm = float(g[case + ".gnorm_core"])
    for k in g.files:
        if k.startswith(case + ".g."):
            pn = k[len(case) + 3:]
            ref = torch.tensor(g[k])
            err = (named[pn].grad.cpu() - ref).norm().item()
            assert err < 2e-4 * ref.norm().item() + 1e-7 * gnorm, (pn, err, ref.norm().item())
    core = fill.interaction_param_shapes(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"])
    tot = torch.sqrt(sum((named[k].grad.double() ** 2).sum() for k in core)).item()
    assert abs(tot - gnorm) / gnorm < 1e-4


def test_all_gradients_against_oracle_autograd():
    """Every core parameter gradient (not just the golden subset) against torch autograd of the oracle."""
    c = fill.ICASES["tiny2"]
    m = build(c).train()
    inp, gi = case_inputs(c)
    x = gi["x"].clone().requires_grad_(True)
    out = m(x, gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
    r = fill.tensor_for("loss.r.oracle", out.shape) * 10.0
    (out * r.to(DEV)).sum().backward()
    p = {k: v.double().requires_grad_(True) for k, v in
         fill.interaction_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"]).items()}
    ref = IR.interaction_forward(p, inp["x"].double(), inp["t"], inp["length"], inp["xf_proj"].double(),
                                 inp["xf_out"].double(), c["H"], c["L"])
    assert rel(out, ref) < 1e-5
    (ref * r.double()).sum().backward()
    named = dict(m.named_parameters())
    for k, v in p.items():
        gr = named[k].grad.cpu().double()
        scale = v.grad.norm().item()
        if k == "sequence_embedding":
            assert gr[c["T"] - 1:].abs().max().item() == 0.0      # rows >= T-1 are never read
        assert (gr - v.grad).norm().item() <= 2e-4 * scale + 1e-5, (k, (gr - v.grad).norm().item(), scale)


@pytest.mark.parametrize("nocross", [False, True])
@pytest.mark.parametrize("case", ["hd64", "hd128"])
def test_bf16_storage_two_person_forward_against_oracle(case, nocross):
    """storage='bf16' on the two-person model (interaction attention with the partner's keys / values and the consumer's
    length, joint_embed2 / out2 on the init-pose rows, 4 stylization blocks per layer): bf16 activations and weights vs
    the fp32 CPU oracle -- bounded at the bf16 level and clearly above fp32 noise."""
    c = {"hd64": dict(B=2, T=33, F=20, d=256, H=4, L=2, ff=256, N=77, Lt=64, num_frames=40, lengths=(33, 12), t=(4, 700)),
         "hd128": dict(B=1, T=61, F=150, d=256, H=2, L=3, ff=512, N=77, Lt=64, num_frames=60, lengths=(45,), t=(250,))}[case]
    m = build(c, no_cross_attn=nocross, storage="bf16").eval()
    inp, gi = case_inputs(c)
    with torch.no_grad():
        out = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
        m.storage = "f32"
        out32 = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
    p = fill.interaction_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"])
    ref = IR.interaction_forward(p, inp["x"], inp["t"], inp["length"], inp["xf_proj"], inp["xf_out"], c["H"], c["L"],
                                 no_cross_attn=nocross)
    assert rel(out32, ref) < 2e-5
    e = rel(out, ref)
    print("two-person bf16 storage %s nocross=%s: rel-L2 vs fp32 oracle %.3e" % (case, nocross, e))
    assert torch.isfinite(out).all() and 1e-4 < e < 3e-2, e
    assert 1e-4 < rel(out[:, 0], ref[:, 0]) < 3e-2          # init-pose rows (joint_embed2 -> ... -> out2)


BF16_TRAIN_CASES = {
    "hd64": dict(B=2, T=33, F=20, d=256, H=4, L=2, ff=256, N=77, Lt=64, num_frames=40, lengths=(33, 12), t=(4, 700)),
    "hd128": dict(B=1, T=61, F=150, d=256, H=2, L=3, ff=512, N=77, Lt=64, num_frames=60, lengths=(45,), t=(250,)),
    # the reference's two-person shape (91 tokens x 263 features, d = 512, 8 heads) on 24 x 91 = 2 184 rows: the
    # weight-stationary GEMMs and the sliced weight-gradient kernel serve the step as they do at full size
    "rows2184": dict(B=12, T=91, F=263, d=512, H=8, L=2, ff=1024, N=77, Lt=256, num_frames=100,
                     lengths=(91, 40, 91, 2, 77, 91, 13, 91, 60, 91, 91, 5), t=(0, 999, 500, 250, 7, 650, 313, 900, 77, 42, 810, 123)),
    # the BENCHMARKED two-person training shape (bench.py `two_person.pit_train_step_ms_bf16_storage`): 32 pairs
    # = 64 model rows x 91 tokens x 263 features, d = 512, 8 heads, L = 8 -- 5 824 rows through every kernel of the bf16 step at
    # its full-size work split, ragged lengths incl. near-empty samples
    "bench64": dict(B=64, T=91, F=263, d=512, H=8, L=8, ff=1024, N=77, Lt=256, num_frames=196,
                    lengths=tuple([91] * 40 + [2 + (11 * i) % 89 for i in range(24)]), t=tuple((37 * i + 5) % 1000 for i in range(64))),
}


@pytest.mark.parametrize("case,nocross", [("hd64", False), ("hd64", True), ("hd128", False), ("rows2184", False), ("bench64", False)])
def test_bf16_storage_two_person_backward_every_gradient_against_oracle_autograd(case, nocross):
    """storage='bf16' TRAINING of the two-person model (what the reference's tools/train.py trains: trainers/mul_ddpm_trainer.py
    :91-163 over models/interaction_transformer.py:167-207,334-367): forward with kept bf16 activations, backward through the
    person <-> person attention (the partner's context matrices, the consumer's length), the joint_embed2 / out2 edges of the
    init-pose rows and the shifted positional table -- EVERY parameter gradient and the three input gradients against torch
    autograd of the fp32 CPU oracle.  Gates as for the single-person model (tests/test_gpu_bf16_training.py): loss 1e-2,
    global gradient norm 2e-2, direction 1 - 1e-3, median tensor 3e-2, no tensor beyond 0.15."""
    c = BF16_TRAIN_CASES[case]
    m = build(c, no_cross_attn=nocross, storage="bf16").train()
    inp, gi = case_inputs(c)
    x, xp, xo = (gi[k].clone().requires_grad_(True) for k in ("x", "xf_proj", "xf_out"))
    out = m(x, gi["t"], length=gi["length"], xf_proj=xp, xf_out=xo)
    target = fill.tensor_for("full.target2", inp["x"].shape) * 10.0
    mask = m.generate_src_mask(c["T"], gi["length"]).to(DEV)
    loss = (((out - target.to(DEV)) ** 2).mean(-1) * mask).sum() / mask.sum()
    loss.backward()
    p = {k: v.clone().requires_grad_(True) for k, v in
         fill.interaction_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"]).items()}
    xr, xpr, xor_ = (inp[k].clone().requires_grad_(True) for k in ("x", "xf_proj", "xf_out"))
    ref = IR.interaction_forward(p, xr, inp["t"], inp["length"], xpr, xor_, c["H"], c["L"], no_cross_attn=nocross)
    ref_loss = (((ref - target) ** 2).mean(-1) * mask.cpu()).sum() / mask.cpu().sum()
    ref_loss.backward()
    named = dict(m.named_parameters())
    e_out = rel(out, ref)
    assert 1e-4 < e_out < 3e-2, e_out
    assert abs(loss.item() - ref_loss.item()) < 1e-2 * abs(ref_loss.item())
    names = [k for k in p if p[k].grad is not None]
    zero_grad = [k for k in names if k.endswith(".key.bias")]          # exactly zero in the reference (softmax over tokens)
    live = [k for k in names if k not in zero_grad]
    assert any("int_ca_block" in k for k in live) != nocross and "joint_embed2.weight" in live and "out2.weight" in live
    for k in live:
        assert named[k].grad is not None and torch.isfinite(named[k].grad).all(), k
    gn = torch.sqrt(sum((named[k].grad.double() ** 2).sum() for k in live)).item()
    gn_ref = torch.sqrt(sum((p[k].grad.double() ** 2).sum() for k in live)).item()
    dot = sum((named[k].grad.double().cpu() * p[k].grad.double()).sum() for k in live).item()
    errs = sorted(((rel(named[k].grad, p[k].grad), k) for k in live), reverse=True)
    print("two-person bf16-storage backward %s nocross=%s: out %.2e, loss %.6f vs %.6f, grad norm %.6e vs %.6e (rel %.2e), cosine %.6f" %
          (case, nocross, e_out, loss.item(), ref_loss.item(), gn, gn_ref, abs(gn - gn_ref) / gn_ref, dot / (gn * gn_ref)))
    print("   worst tensors: " + ", ".join("%s %.2e" % (k, e) for e, k in errs[:5]))
    print("   median rel-L2 over %d tensors: %.2e; joint_embed2.weight %.2e, out2.weight %.2e, sequence_embedding %.2e" %
          (len(errs), errs[len(errs) // 2][0], rel(named["joint_embed2.weight"].grad, p["joint_embed2.weight"].grad),
           rel(named["out2.weight"].grad, p["out2.weight"].grad), rel(named["sequence_embedding"].grad, p["sequence_embedding"].grad)))
    assert abs(gn - gn_ref) < 2e-2 * gn_ref, (gn, gn_ref)
    assert dot / (gn * gn_ref) > 1 - 1e-3
    assert errs[0][0] < 0.15, errs[0]
    assert errs[len(errs) // 2][0] < 3e-2
    if c["num_frames"] > c["T"] - 1:
        assert named["sequence_embedding"].grad[c["T"] - 1:].abs().max().item() == 0.0  # rows >= T - 1 are never read
    for a, b in ((x.grad, xr.grad), (xp.grad, xpr.grad), (xo.grad, xor_.grad)):
        assert rel(a, b) < 5e-2
    for k in zero_grad:
        assert named[k].grad.norm().item() < 2e-2 * gn_ref, k


@pytest.mark.parametrize("with_label", [False, True])
def test_fused_two_person_step_with_bf16_storage_tracks_the_fp32_step(with_label):
    """DDPMMulTrainer.train_step_fused with storage='bf16' (PIT and labelled mode) next to the same steps with fp32 storage:
    loss within 1e-2, clipped-gradient norm within 2e-2 over three steps; the bf16 weight shadow equals bf16(master) after
    every step; the hipGraph-captured step replays the eager one bit for bit."""
    c = BF16_TRAIN_CASES["hd64"]
    B, T, Fd = c["B"], c["T"], c["F"]
    rows = 2 * B if with_label else 4 * B
    x0 = (fill.tensor_for("fused2b.x0", (2 * B, T, Fd)) * 10).to(DEV)
    g = torch.Generator().manual_seed(11)
    noises = [torch.randn(x0.shape, generator=g).to(DEV) for _ in range(3)]
    tt = torch.tensor(c["t"], device=DEV)
    length = torch.tensor(c["lengths"], device=DEV)
    xf_proj = (fill.tensor_for("fused2b.xp", (rows, 4 * c["d"])) * 10).to(DEV)
    xf_out = (fill.tensor_for("fused2b.xo", (rows, c["N"], c["Lt"])) * 10).to(DEV)
    label = "labels/" if with_label else None
    runs = {}
    for storage in ("f32", "bf16"):
        m = build(c, storage=storage).train()
        tr = _trainer(c, m, label_path=label)
        losses, gns = [], []
        for k in range(3):
            l = tr.train_step_fused(x0, tt, length, xf_proj, xf_out, noise=noises[k])
            losses.append(l.item())
            gns.append(tr.fused_state()["gnorm"].item())
            if storage == "bf16":
                fp = m.flat_params()
                sh = fp.shadow16_buffer()
                assert torch.equal(sh[:fp.core_numel], fp.flat[:fp.core_numel].to(torch.bfloat16))
        runs[storage] = (losses, gns)
    print("two-person fused step, with_label=%s: losses f32 %s bf16 %s, gnorm f32 %s bf16 %s" % (with_label, *[
        ["%.5f" % v for v in runs[s][i]] for i in (0, 1) for s in ("f32", "bf16")]))
    for a, b in zip(runs["f32"][0], runs["bf16"][0]):
        assert abs(a - b) < 1e-2 * abs(a)
    for a, b in zip(runs["f32"][1], runs["bf16"][1]):
        assert abs(a - b) < 2e-2 * abs(a)
    # captured == eager, bit for bit
    outs = []
    for captured in (False, True):
        m = build(c, storage="bf16").train()
        tr = _trainer(c, m, label_path=label)
        step = tr.train_step_captured if captured else tr.train_step_fused
        ls = [step(x0, tt, length, xf_proj, xf_out, noise=noises[k]).item() for k in range(2)]
        outs.append((ls, {k: v.clone() for k, v in m.state_dict().items()}))
    assert outs[0][0] == outs[1][0]
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


@pytest.mark.parametrize("storage", ["bf16", "f32"])
def test_captured_pit_step_does_not_depend_on_when_the_host_synchronises(storage):
    """Regression (round 5): at bench.py's two-person size the captured bf16 PIT step diverged (loss 1.97 -> 2.6 -> 3.3 ... -> NaN) when the
    graph was replayed onto an IDLE device -- two warm-up steps, torch.cuda.synchronize(), then steps back to back, i.e. exactly
    bench.py's timed() -- and trained normally without the synchronisation: a hipMemset2DAsync / hipMemcpyAsync pair of the
    two-person bf16 backward (init-pose rows, shifted positional table) ran out of order as graph nodes.  The fp32 step had the same
    hazard (hipMemcpyAsync nodes), hidden as long as its weight gradients forked onto a second stream under capture.  The library
    launches kernels for every fill / copy now; the loss trajectory must not depend on the synchronisation pattern."""
    import hig_amd.trainers  # noqa: F401
    c = dict(B=64, T=91, F=263, d=512, H=8, L=8, ff=1024, N=77, Lt=256, num_frames=196)
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(32, c["T"], c["F"], generator=g).to(DEV)
    nz = torch.randn(32, c["T"], c["F"], generator=g).to(DEV)
    tt = torch.randint(0, 1000, (16,), generator=g).to(DEV)
    ln = torch.randint(20, c["T"], (16,), generator=g).to(DEV)
    xp = torch.randn(64, 4 * c["d"], generator=g).to(DEV)
    xo = torch.randn(64, c["N"], c["Lt"], generator=g).to(DEV)
    runs = []
    for sync_after in (None, 1):
        m = build(c, storage=storage).train()
        tr = _trainer(dict(c, B=16), m)
        losses = []
        for k in range(6):
            tr.train_step_captured(x0, tt, ln, xp, xo, noise=nz)
            losses.append(tr.fused_state()["loss"].clone())
            if sync_after is not None and k == sync_after:
                torch.cuda.synchronize()
        runs.append([v.item() for v in losses])
        del tr, m
    assert runs[0] == runs[1], runs
    assert all(v == v and abs(v) < float("inf") for v in runs[0]), runs[0]


def test_swapping_the_two_persons_swaps_the_outputs():
    """Size-independent property at a production-like size: the model is symmetric in the two
    persons, so model(cat[x2, x1]) == swap(model(cat[x1, x2])) when both share text / t / length."""
    c = dict(B=16, T=100, F=150, d=512, H=8, L=4, ff=1024, N=77, Lt=256, num_frames=196)
    m = build(c).eval()
    B = c["B"]
    lengths = [c["T"] - (7 * i) % 60 for i in range(B)]
    inp = fill.inputs(2 * B, c["T"], c["F"], c["d"], c["N"], c["Lt"], lengths * 2, [(37 * i) % 1000 for i in range(B)] * 2)
    gi = {k: v.to(DEV) for k, v in inp.items()}
    for k in ("xf_proj", "xf_out"):      # same conditioning for both persons of a pair
        gi[k] = torch.cat([gi[k][:B], gi[k][:B]])
    with torch.no_grad():
        a = m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
        xs = torch.cat([gi["x"][B:], gi["x"][:B]])
        b = m(xs, gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
    assert torch.isfinite(a).all() and a.abs().max() > 0
    assert rel(torch.cat([b[B:], b[:B]]), a) < 1e-6
    # ...and it is a real interaction: changing person 2 changes person 1's output
    with torch.no_grad():
        x2 = gi["x"].clone()
        x2[B:] += 1.0
        c2 = m(x2, gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
    assert rel(c2[:B], a[:B]) > 1e-4
    # against the oracle on the first pairs of the same batch (pairs are independent)
    p = fill.interaction_params(c["F"], c["d"], c["ff"], c["L"], c["Lt"], c["num_frames"])
    sel = [0, 1, B, B + 1]
    ref = IR.interaction_forward(p, inp["x"][sel], inp["t"][sel], inp["length"][sel], gi["xf_proj"][sel].cpu(),
                                 gi["xf_out"][sel].cpu(), c["H"], c["L"])
    assert rel(a[sel], ref) < 2e-5


def test_split_bf16_products_keep_parity(gold):
    g = gold("g8_interaction.npz")
    c = fill.ICASES["config1x2"]
    m = build(c, precision="bf16x3").train()
    _, gi = case_inputs(c)
    x = gi["x"].clone().requires_grad_(True)
    out = m(x, gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
    assert rel(out, g["config1x2.out"]) < 5e-5
    r = (fill.tensor_for("loss.r.config1x2", out.shape) * 10.0).to(DEV)
    (out * r).sum().backward()
    assert rel(x.grad, g["config1x2.dx"]) < 2e-4


def _trainer(c, m, label_path=None):
    args = types.SimpleNamespace(device=torch.device(DEV), diffusion_steps=1000, is_train=True, lr=2e-4,
                                 batch_size=c["B"], num_epochs=1, log_every=50, save_latest=500, save_every_e=5,
                                 is_continue=False, model_dir="/tmp", multi=True, label_path=label_path, cap_id=False)
    return hig_amd.DDPMMulTrainer(args, m)


CAP1 = ["a person shakes hands with another person", "one pushes the other"]
CAP2 = ["a person receives a handshake", "one is pushed by the other"]


def test_pit_trainer_step_matches_reference_golden(gold):
    """DDPMMulTrainer.forward + update in PIT mode (forward_twice, min over caption assignments) == G9."""
    import hig_amd.trainers.ddpm_trainer as tr
    g = gold("g9_mul_trainer.npz")
    c = fill.ICASES["config1x2"]
    m = build(c).train()
    trainer = _trainer(c, m)
    trainer.opt_encoder = torch.optim.Adam(m.parameters(), lr=2e-4)
    B, T, Fd = c["B"], c["T"], c["F"]
    motion1 = fill.tensor_for("g9.motion1", (B, T, Fd)) * 10
    motion2 = fill.tensor_for("g9.motion2", (B, T, Fd)) * 10
    t_fixed = torch.tensor(c["t"])
    trainer.sampler.sample = lambda bs, dev: (t_fixed.to(dev), torch.ones(bs))
    captured = {}
    real_clip = tr.clip_grad_norm_

    def spy(params, max_norm):
        captured["gnorm"] = float(real_clip(params, max_norm))
        return captured["gnorm"]

    tr.clip_grad_norm_ = spy
    undo = _patch(_NoiseFeed("g9.noise", DEV))
    try:
        trainer.forward((CAP1, CAP2, motion1, motion2, torch.tensor(c["lengths"]), None))
        logs = trainer.update()
    finally:
        undo()
        tr.clip_grad_norm_ = real_clip
    assert trainer.fake_noise.shape == (4 * B, T, Fd)
    assert rel(trainer.fake_noise, g["fake_noise"]) < 5e-5
    assert np.array_equal(trainer.src_mask.cpu().numpy(), g["src_mask"])
    assert abs(logs["loss_mot_rec"] - float(g["loss_mot_rec"])) < 1e-4 * float(g["loss_mot_rec"])
    assert abs(captured["gnorm"] - float(g["gnorm"])) < 2e-4 * float(g["gnorm"])
    sd = m.state_dict()
    for k in g.files:
        if k.startswith("p."):
            assert (sd[k[2:]].cpu() - torch.tensor(g[k])).abs().max().item() < 2e-5, k


def test_labelled_trainer_loss_and_generate():
    """with_label branch (no forward_twice) against the oracle's loss; then a short sampling run."""
    c = fill.ICASES["config1x2"]
    m = build(c).train()
    trainer = _trainer(c, m, label_path="labels/")
    trainer.opt_encoder = torch.optim.Adam(m.parameters(), lr=2e-4)
    B, T, Fd = c["B"], c["T"], c["F"]
    motion1 = fill.tensor_for("lab.motion1", (B, T, Fd)) * 10
    motion2 = fill.tensor_for("lab.motion2", (B, T, Fd)) * 10
    trainer.forward((CAP1, CAP2, motion1, motion2, torch.tensor(c["lengths"]), None))
    assert trainer.fake_noise.shape == (2 * B, T, Fd)
    logs = trainer.update()
    ref = IR.labelled_loss(trainer.fake_noise.detach().cpu(), trainer.real_noise.cpu(), trainer.src_mask.cpu())
    assert abs(logs["loss_mot_rec"] - ref.item()) < 1e-5 * ref.item()
    # sampling: 50-step schedule, hipGraph loop, both persons come back as (motion1, motion2) pairs
    trainer.diffusion = make_diffusion(50)
    outs = trainer.generate(CAP1, CAP2, torch.tensor([T, 40]), Fd)
    Tg = min(T, c["num_frames"])
    assert len(outs) == B and all(len(o) == 2 and o[0].shape == (Tg, Fd) for o in outs)
    assert all(torch.isfinite(o[0]).all() and torch.isfinite(o[1]).all() for o in outs)


def test_two_person_generate_with_bf16_storage():
    """DDPMMulTrainer.generate (captured p_sample_loop over the two-person model) with storage='bf16': runs, is finite,
    returns (motion1, motion2) pairs, and repeats itself bitwise under the same noise stream."""
    c = dict(B=2, T=33, F=20, d=256, H=4, L=2, ff=256, N=77, Lt=64, num_frames=40, lengths=(33, 12), t=(4, 700))
    m = build(c, storage="bf16").eval()
    trainer = _trainer(c, m)
    trainer.diffusion = make_diffusion(30)
    outs = []
    for _ in range(2):
        torch.manual_seed(7)
        outs.append(trainer.generate(CAP1[:2], CAP2[:2], torch.tensor([33, 20]), c["F"]))
    assert len(outs[0]) == 2 and all(len(o) == 2 and o[0].shape == (33, c["F"]) for o in outs[0])
    for a, b in zip(outs[0], outs[1]):
        assert torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_cap_id_mode_class_embeddings_single_text_token():
    """cap_id=True: captions are class ids, the text context is ONE token per sample (N=1,
    interaction_transformer.py:558-563); forward + backward against the oracle."""
    c = fill.ICASES["tiny2"]
    m = build(c, cap_id=True).train()
    assert not hasattr(m, "clip") and m.cap_embedding.shape == (43, c["Lt"])
    inp, gi = case_inputs(c)
    ids = [torch.tensor([3, 41], device=DEV), torch.tensor([4, 40], device=DEV)]
    x = gi["x"].clone().requires_grad_(True)
    out = m(x, gi["t"], length=gi["length"], text=ids)
    r = fill.tensor_for("loss.r.capid", out.shape) * 10.0
    (out * r.to(DEV)).sum().backward()
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in m.state_dict().items()}
    emb = sd["cap_embedding"][torch.cat(ids).cpu()]
    xf_proj = torch.nn.functional.linear(emb, sd["text_proj.0.weight"], sd["text_proj.0.bias"])
    xr = inp["x"].double().requires_grad_(True)
    ref = IR.interaction_forward(sd, xr, inp["t"], inp["length"], xf_proj, emb.unsqueeze(1), c["H"], c["L"])
    assert rel(out, ref) < 1e-5
    (ref * r.double()).sum().backward()
    assert rel(x.grad, xr.grad) < 1e-4
    assert rel(m.cap_embedding.grad, sd["cap_embedding"].grad) < 1e-4          # through xf_out AND xf_proj
    assert rel(m.text_proj[0].weight.grad, sd["text_proj.0.weight"].grad) < 1e-4


@pytest.mark.parametrize("with_label", [False, True])
def test_fused_two_person_step_equals_reference_sequence(with_label):
    """DDPMMulTrainer.train_step_fused (hig_pair_mse + flat clip+Adam) == training_losses(forward_twice) +
    backward_G + clip + Adam through autograd, on the core parameters; then the captured step == the fused one."""
    c = fill.ICASES["config1x2"]
    B, T, Fd = c["B"], c["T"], c["F"]
    rows = 2 * B if with_label else 4 * B
    x0 = (fill.tensor_for("fused2.x0", (2 * B, T, Fd)) * 10).to(DEV)
    noise = (fill.tensor_for("fused2.noise", x0.shape) * 10).to(DEV)
    tt = torch.tensor(c["t"], device=DEV)
    length = torch.tensor(c["lengths"], device=DEV)
    xf_proj = (fill.tensor_for("fused2.xp", (rows, 4 * c["d"])) * 10).to(DEV)
    xf_out = (fill.tensor_for("fused2.xo", (rows, c["N"], c["Lt"])) * 10).to(DEV)
    label = "labels/" if with_label else None
    # reference sequence through autograd, core parameters only
    m1 = build(c).train()
    tr1 = _trainer(c, m1, label_path=label)
    opt = torch.optim.Adam(m1.core_parameters(), lr=2e-4)
    len_rows = torch.cat([length] * (rows // B))
    out = tr1.diffusion.training_losses(m1, x0, torch.cat([tt, tt]), noise=noise, forward_twice=not with_label,
                                        model_kwargs={"xf_proj": xf_proj, "xf_out": xf_out, "length": len_rows})
    tr1.real_noise, tr1.fake_noise = out["target"], out["pred"]
    tr1.src_mask = m1.generate_src_mask(T, len_rows).to(DEV)
    tr1.backward_G()
    tr1.loss_mot_rec.backward()
    gn = torch.nn.utils.clip_grad_norm_(m1.core_parameters(), 0.5)
    opt.step()
    oracle = (IR.labelled_loss if with_label else IR.pit_loss)(out["pred"].detach().cpu(), out["target"].cpu(),
                                                              tr1.src_mask.cpu())
    assert abs(tr1.loss_mot_rec.item() - oracle.item()) < 1e-5 * oracle.item()
    # fused
    m2 = build(c).train()
    tr2 = _trainer(c, m2, label_path=label)
    l2 = tr2.train_step_fused(x0, tt, length, xf_proj, xf_out, noise=noise)
    st = tr2.fused_state()
    assert abs(l2.item() - tr1.loss_mot_rec.item()) < 1e-5 * abs(l2.item())
    assert abs(st["gnorm"].item() - gn.item()) < 1e-4 * gn.item()
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        # Adam's first step is lr * g / (|g| + eps): elements whose gradient is within rounding of 1e-8 move by a
        # visibly different fraction of lr = 2e-4 (key.bias: its true gradient is zero, pure rounding noise)
        assert (a - b).abs().max().item() < 5e-6, k
    # captured
    m3 = build(c).train()
    tr3 = _trainer(c, m3, label_path=label)
    l3 = tr3.train_step_captured(x0, tt, length, xf_proj, xf_out, noise=noise)
    assert abs(l3.item() - l2.item()) <= 1e-6 * abs(l2.item())
    for (k, a), (_, b) in zip(m2.state_dict().items(), m3.state_dict().items()):
        assert torch.equal(a, b), k


def test_pit_fused_batch_with_text_head_matches_reference_golden(gold):
    """train_fused_batch (PIT mode, text head inside the captured step) reproduces the reference's own
    DDPMMulTrainer.forward + update (golden g9): loss, pre-clip gradient norm, updated parameters."""
    g = gold("g9_mul_trainer.npz")
    c = fill.ICASES["config1x2"]
    m = build(c).train()
    trainer = _trainer(c, m)
    B, T, Fd = c["B"], c["T"], c["F"]
    motion1 = fill.tensor_for("g9.motion1", (B, T, Fd)) * 10
    motion2 = fill.tensor_for("g9.motion2", (B, T, Fd)) * 10
    t_fixed = torch.tensor(c["t"])
    trainer.sampler.sample = lambda bs, dev: (t_fixed.to(dev), torch.ones(bs))
    noise = (fill.tensor_for("g9.noise.0", (2 * B, T, Fd)) * 10).to(DEV)
    loss = trainer.train_fused_batch((CAP1, CAP2, motion1, motion2, torch.tensor(c["lengths"]), None), captured=True,
                                     noise=noise)
    st = trainer.fused_state()
    assert abs(loss.item() - float(g["loss_mot_rec"])) < 1e-4 * float(g["loss_mot_rec"])
    assert abs(st["gnorm"].item() - float(g["gnorm"])) < 2e-4 * float(g["gnorm"])
    sd = m.state_dict()
    for k in g.files:
        if k.startswith("p."):
            assert (sd[k[2:]].cpu() - torch.tensor(g[k])).abs().max().item() < 2e-5, k
