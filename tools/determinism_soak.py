"""Race / determinism soak (tuning + QA aid): repeats the same forward + backward (+ text head) many times and checks
that every output and the whole flat gradient buffer are BITWISE identical each time (no float atomics anywhere, so any
difference would be a missing barrier or an uninitialised read)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, hig_amd
dev = torch.device("cuda", 0)
n = int(os.environ.get("ITERS", 40))

def soak(name, m, i, two=False):
    m.train()
    caps = ["a person walks towards another person number %d" % k for k in range(i["x"].shape[0])]
    tok, feat = m._clip_features(caps, dev)
    ref = None
    for it in range(n):
        m.zero_grad(set_to_none=True)
        xp, xo = m._text_head(tok, feat)
        out, saved = m._launch_forward(i["x"], i["t"], i["length"], xp.detach(), xo.detach(), training=True)
        dx, dxp, dxo = m._launch_backward(i["x"], i["t"], i["length"], xo.detach(), saved, i["x0"], want_dx=True)
        (xp * dxp).sum().backward(retain_graph=True)
        tg = torch.cat([p.grad.reshape(-1) for p in m._text_params()])
        cur = (out.clone(), dx.clone(), dxp.clone(), dxo.clone(), m.flat_params().grad[:m.flat_params().core_numel].clone(), tg.clone())
        if ref is None:
            ref = cur
            assert all(torch.isfinite(t).all() for t in cur)
        else:
            for k, (a, b) in enumerate(zip(ref, cur)):
                assert torch.equal(a, b), (name, it, k, (a - b).abs().max().item())
    print("%s: %d identical iterations" % (name, n))

c = dict(bench.CFG)
soak("single-person config 2", bench.build_model(c, dev), bench.make_inputs(c, dev, 0))
c2 = dict(c, B=64, T=91, F=263)
torch.manual_seed(0)
m2 = hig_amd.MotionInteractionTransformer(input_feats=c2["F"], num_frames=196, latent_dim=c2["d"], ff_size=c2["ff"],
                                          num_layers=c2["L"], num_heads=c2["H"], text_latent_dim=c2["Lt"])
with torch.no_grad():
    for name, p in m2.named_parameters():
        if name.startswith("out") or ".ffn.linear2." in name or ".out_layers.2." in name:
            p.copy_(torch.randn(p.shape) * 0.02)
i2 = bench.make_inputs(c2, dev, 0)
i2["length"] = (torch.arange(64, device=dev) * 7 % 92).long()       # ragged, incl. an empty sample
soak("two-person, ragged lengths", m2.to(dev), i2)

# bf16 storage (round 4): the training step's own kernels (wgrad16 on the side stream, bf16 row / attention backward, the F-wide
# edges) and the per-call inference forward with its B-row work on the library's third stream, at config 2 and at the
# config-5 width (LayerNorm fold at K = 1024: one staging buffer in the K-split GEMM)
def soak16(name, c, layers=None):
    cc = dict(c) if layers is None else dict(c, L=layers)
    m = bench.build_model(cc, dev)
    bench.set_mode(m, "bf16s")
    i = bench.make_inputs(cc, dev, 0)
    i["length"] = (torch.arange(cc["B"], device=dev) * 37 % cc["T"] + 1).long()
    m.train()
    ref = None
    for it in range(n):
        m.zero_grad(set_to_none=True)
        out, saved = m._launch_forward(i["x"], i["t"], i["length"], i["xf_proj"], i["xf_out"], training=True)
        dx, dxp, dxo = m._launch_backward(i["x"], i["t"], i["length"], i["xf_out"], saved, i["x0"], want_dx=True)
        cur = [out.clone(), dx.clone(), dxp.clone(), dxo.clone(), m.flat_params().grad[:m.flat_params().core_numel].clone()]
        m.eval(); m.cache_text_context = False
        with torch.no_grad():
            cur.append(m(i["x"], i["t"], length=i["length"], xf_proj=i["xf_proj"], xf_out=i["xf_out"]).clone())
        m.train()
        if ref is None:
            ref = cur
            assert all(torch.isfinite(t.float()).all() for t in cur)
        else:
            for k, (a, b) in enumerate(zip(ref, cur)):
                assert torch.equal(a, b), (name, it, k, (a.float() - b.float()).abs().max().item())
    print("%s: %d identical iterations" % (name, n))

# round 6: the fp32 per-call INFERENCE forward -- gemm_wsp32 / the batched text side (one K = 256 GEMM, one context build over L H
# heads) / apply_wave64 with its opaque prefetch loads and counted waits -- with ragged lengths, back to back
def soak32_inference(name, c):
    m = bench.build_model(c, dev).eval()
    m.cache_text_context = False
    i = bench.make_inputs(c, dev, 0)
    i["length"] = (torch.arange(c["B"], device=dev) * 37 % c["T"] + 1).long()
    ref = None
    with torch.no_grad():
        for it in range(n):
            out = m(i["x"], i["t"], length=i["length"], xf_proj=i["xf_proj"], xf_out=i["xf_out"]).clone()
            if ref is None:
                ref = out
                assert torch.isfinite(out).all()
            else:
                assert torch.equal(out, ref), (name, it, (out - ref).abs().max().item())
    print("%s: %d identical iterations" % (name, n))

soak32_inference("fp32 per-call inference forward, config 2", dict(bench.CFG))
soak16("bf16 storage, config 2 (training step + per-call inference forward)", bench.CFG)
soak16("bf16 storage, config-5 width (B=32 T=300 d=1024 H=8, 3 layers)", dict(bench.CFG, B=32, T=300, d=1024, H=8, ff=1024), layers=3)

# round 5: the two-person model with bf16 storage (person <-> person block, grouped weight-gradient launches, batched reductions,
# kernels where graph memset / memcpy nodes used to be), forward + backward + the captured PIT step replayed from the same state
def soak16_pair(name):
    import types
    c2 = dict(bench.CFG, B=64, T=91, F=263)
    torch.manual_seed(0)
    m = hig_amd.MotionInteractionTransformer(input_feats=c2["F"], num_frames=196, latent_dim=c2["d"], ff_size=c2["ff"], num_layers=c2["L"],
                                             num_heads=c2["H"], text_latent_dim=c2["Lt"], storage="bf16")
    with torch.no_grad():
        for pname, p in m.named_parameters():
            if pname.startswith("out") or ".ffn.linear2." in pname or ".out_layers.2." in pname: p.copy_(torch.randn(p.shape) * 0.02)
    m = m.to(dev).train()
    i = bench.make_inputs(c2, dev, 0)
    i["length"] = (torch.arange(c2["B"], device=dev) * 37 % (c2["T"] - 1) + 1).long()
    ref = None
    for it in range(n):
        m.zero_grad(set_to_none=True)
        out, saved = m._launch_forward(i["x"], i["t"], i["length"], i["xf_proj"], i["xf_out"], training=True)
        dx, dxp, dxo = m._launch_backward(i["x"], i["t"], i["length"], i["xf_out"], saved, i["x0"], want_dx=True)
        cur = [out.clone(), dx.clone(), dxp.clone(), dxo.clone(), m.flat_params().grad[:m.flat_params().core_numel].clone()]
        if ref is None:
            ref = cur
            assert all(torch.isfinite(t.float()).all() for t in cur)
        else:
            for k, (a, b) in enumerate(zip(ref, cur)):
                assert torch.equal(a, b), (name, it, k, (a.float() - b.float()).abs().max().item())
    print("%s: %d identical iterations" % (name, n))

soak16_pair("two-person model, bf16 storage (forward + backward, 64 rows x 91 tokens x 263 features)")
