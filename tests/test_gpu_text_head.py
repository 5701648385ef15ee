"""Text head on the MI355X: hig_text_head_fwd/_bwd behind MotionTransformer.encode_text
against the reference's own encode_text (golden g10), the CPU oracle and the stock-torch head."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import hig_amd  # noqa: E402
from oracle import fill  # noqa: E402
from oracle import text_head_ref as TH  # noqa: E402
from test_gpu_denoiser import rel  # noqa: E402
from test_oracle_golden import TEXT_CAPTIONS  # noqa: E402

DEV = "cuda"


def build(c, **kw):
    m = hig_amd.MotionTransformer(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"],
                                  ff_size=c["ff"], num_layers=c["L"], num_heads=c["H"],
                                  text_latent_dim=c["Lt"], **kw)
    m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
    return m.to(DEV)


@pytest.mark.parametrize("case", ["tiny", "config1"])
def test_encode_text_matches_reference_golden(gold, case):
    g = gold("g10_text_head.npz")
    c = fill.CASES[case]
    m = build(c).train()
    assert m.text_head == "hip"
    xf_proj, xf_out = m.encode_text(TEXT_CAPTIONS, DEV)
    assert xf_proj.shape == (4, 4 * c["d"]) and xf_out.shape == (4, 77, c["Lt"])
    assert rel(xf_proj, g[case + ".xf_proj"]) < 1e-5 and rel(xf_out, g[case + ".xf_out"]) < 1e-5
    r1 = (fill.tensor_for("g10.r1." + case, xf_proj.shape) * 10.0).to(DEV)
    r2 = (fill.tensor_for("g10.r2." + case, xf_out.shape) * 10.0).to(DEV)
    ((xf_proj * r1).sum() + (xf_out * r2).sum()).backward()
    named = dict(m.named_parameters())
    tot = torch.sqrt(sum((v.grad.double() ** 2).sum() for k, v in named.items() if k.startswith("text"))).item()
    gn = float(g[case + ".gnorm_text"])
    assert abs(tot - gn) < 1e-4 * gn
    for k in g.files:
        if k.startswith(case + ".g."):
            ref = torch.tensor(g[k])
            got = named[k[len(case) + 3:]].grad.cpu()
            if got.shape != ref.shape:
                got = got[:64, :64]
            assert (got - ref).norm().item() < 2e-4 * ref.norm().item() + 1e-7 * gn, k
    assert all(p.grad is None for p in m.clip.parameters())              # CLIP stays frozen


def test_every_text_gradient_against_oracle_and_inference_path():
    c = fill.CASES["config1"]
    m = build(c).train()
    tokens, x = m._clip_features(TEXT_CAPTIONS, DEV)
    xf_proj, xf_out = m._text_head(tokens, x)
    with torch.no_grad():                                               # inference path: ping-pong buffers
        ip, io = m._text_head(tokens, x)
    assert torch.equal(ip, xf_proj) and torch.equal(io, xf_out)
    r1 = fill.tensor_for("th.r1", xf_proj.shape) * 10.0
    r2 = fill.tensor_for("th.r2", xf_out.shape) * 10.0
    ((xf_proj * r1.to(DEV)).sum() + (xf_out * r2.to(DEV)).sum()).backward()
    p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in m.state_dict().items() if k.startswith("text")}
    clip_out = x.permute(1, 0, 2).cpu().double()
    op, oo = TH.text_head_forward(p, clip_out, tokens.argmax(-1).cpu(), 4, 4)
    assert rel(xf_proj, op) < 1e-5 and rel(xf_out, oo) < 1e-5
    ((op * r1.double()).sum() + (oo * r2.double()).sum()).backward()
    named = dict(m.named_parameters())
    for k, v in p.items():
        got = named[k].grad.cpu().double()
        assert (got - v.grad).norm().item() < 2e-4 * v.grad.norm().item() + 1e-6, k
    # only-xf_proj upstream (xf_out unused downstream): the other gradient arrives as None
    m.zero_grad()
    a, _ = m._text_head(tokens, x)
    a.sum().backward()
    assert named["text_proj.0.bias"].grad.abs().sum() > 0


def test_hip_text_head_equals_stock_torch_head_in_training_step():
    """Same trainer step with text_head='hip' and 'torch': losses and updated text parameters agree."""
    import types
    c = fill.CASES["config1"]
    res = {}
    for mode in ("hip", "torch"):
        m = build(c, text_head=mode).train()
        args = types.SimpleNamespace(device=torch.device(DEV), diffusion_steps=1000, is_train=True, lr=2e-4,
                                     batch_size=c["B"], num_epochs=1, log_every=50, save_latest=500, save_every_e=5,
                                     is_continue=False, model_dir="/tmp")
        tr = hig_amd.DDPMTrainer(args, m)
        tr.opt_encoder = torch.optim.Adam(m.parameters(), lr=2e-4)
        torch.manual_seed(3)
        tr.sampler.sample = lambda bs, dev: (torch.tensor(c["t"]).to(dev), torch.ones(bs))
        motions = fill.tensor_for("th.motions", (c["B"], c["T"], c["F"])) * 10
        tr.forward((TEXT_CAPTIONS[:2], motions, torch.tensor(c["lengths"])))
        logs = tr.update()
        res[mode] = (logs["loss_mot_rec"], {k: v.detach().cpu().clone() for k, v in m.state_dict().items()
                                            if k.startswith("text")})
    assert abs(res["hip"][0] - res["torch"][0]) < 1e-5 * abs(res["torch"][0])
    for k, v in res["torch"][1].items():
        assert (res["hip"][1][k] - v).abs().max().item() < 2e-5, k


def test_identity_pre_proj_and_split_bf16():
    """text_latent_dim=512 makes text_pre_proj an nn.Identity (transformer.py:325-328) and the head dim 128 (matrix-core
    attention kernels): HIP head == torch head (text_head='torch').  Lt=256 + bf16x3 stays in parity."""
    c = dict(fill.CASES["tiny"], Lt=512)
    m = build(c)
    got = m.encode_text(TEXT_CAPTIONS[:2], DEV)
    m.text_head = "torch"
    a, b = m.encode_text(TEXT_CAPTIONS[:2], DEV)
    assert a.shape == (2, 4 * c["d"]) and b.shape == (2, 77, 512)
    assert rel(got[0], a) < 2e-5 and rel(got[1], b) < 2e-5
    c = fill.CASES["config1"]
    m = build(c)
    ref = m.encode_text(TEXT_CAPTIONS, DEV)
    m.precision = "bf16x3"
    got = m.encode_text(TEXT_CAPTIONS, DEV)
    assert rel(got[0], ref[0]) < 5e-5 and rel(got[1], ref[1]) < 5e-5


def test_fused_step_with_text_head_equals_full_reference_update():
    """train_fused_batch (text head inside the fused / captured step, one clip+Adam over core + text parameters)
    == the reference sequence forward() + update() with torch's Adam over EVERY trainable parameter."""
    import types
    import hig_amd.models.gaussian_diffusion as gdm
    from test_gpu_denoiser import _NoiseFeed, _patch
    c = fill.CASES["config1"]
    motions = fill.tensor_for("fts.motions", (c["B"], c["T"], c["F"])) * 10
    batch = (TEXT_CAPTIONS[:2], motions, torch.tensor(c["lengths"]))
    t_fixed = torch.tensor(c["t"])
    res = {}
    for mode in ("reference", "fused", "captured"):
        m = build(c).train()
        args = types.SimpleNamespace(device=torch.device(DEV), diffusion_steps=1000, is_train=True, lr=2e-4,
                                     batch_size=c["B"], num_epochs=1, log_every=50, save_latest=500, save_every_e=5,
                                     is_continue=False, model_dir="/tmp")
        tr = hig_amd.DDPMTrainer(args, m)
        tr.sampler.sample = lambda bs, dev: (t_fixed.to(dev), torch.ones(bs))
        noise = (fill.tensor_for("fts.noise", motions.shape) * 10).to(DEV)
        if mode == "reference":
            tr.opt_encoder = torch.optim.Adam(m.parameters(), lr=2e-4)
            undo = _patch(type("F", (), {"randn": None, "randn_like": staticmethod(lambda x, **_: noise)})())
            try:
                tr.forward(batch)
                loss = tr.update()["loss_mot_rec"]
            finally:
                undo()
        else:
            loss = tr.train_fused_batch(batch, captured=(mode == "captured"), noise=noise).item()
            fp = m.flat_params()
            assert len(fp.text_params) == 6 + 12 * 4 and fp.numel > fp.core_numel
        res[mode] = (loss, {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if not k.startswith("clip.")})
    for mode in ("fused", "captured"):
        assert abs(res[mode][0] - res["reference"][0]) < 1e-5 * abs(res["reference"][0]), mode
        moved = 0
        for k, v in res["reference"][1].items():
            # Adam's first step moves by lr * g / (|g| + eps): only gradients within rounding of 1e-8 can differ visibly
            assert (res[mode][1][k] - v).abs().max().item() < 5e-6, (mode, k)
            moved += int(k.startswith("text") and (v - fill.tensor_for(k, v.shape)).abs().max().item() > 1e-5)
        assert moved >= 50, moved              # the text head really was trained
    for k, v in res["fused"][1].items():
        assert torch.equal(res["captured"][1][k], v), k
