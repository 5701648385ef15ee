"""bench.py's two-person section in isolation (diagnosis of a NaN loss of the bf16 PIT step that only the bench sequence showed):
fp32 inference forward, forward + backward, fp32 PIT steps (captured), then a bf16-storage copy of the TRAINED model and its steps."""
import os, sys, types, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, hig_amd
dev = torch.device("cuda", 0)
skip = set(os.environ.get("SKIP", "").split(","))
c2 = dict(bench.CFG, B=64, T=91, F=263)
torch.manual_seed(0)
m2 = hig_amd.MotionInteractionTransformer(input_feats=c2["F"], num_frames=196, latent_dim=c2["d"], ff_size=c2["ff"], num_layers=c2["L"],
                                          num_heads=c2["H"], text_latent_dim=c2["Lt"])
with torch.no_grad():
    for name, p in m2.named_parameters():
        if name.startswith("out") or ".ffn.linear2." in name or ".out_layers.2." in name:
            p.copy_(torch.randn(p.shape) * 0.02)
m2 = m2.to(dev)
i2 = bench.make_inputs(c2, dev, 0)
if "fwd" not in skip:
    with torch.no_grad():
        for _ in range(12): m2(i2["x"], i2["t"], length=i2["length"], xf_proj=i2["xf_proj"], xf_out=i2["xf_out"])
if "fwdbwd" not in skip:
    for _ in range(12):
        out, saved = m2._launch_forward(i2["x"], i2["t"], i2["length"], i2["xf_proj"], i2["xf_out"], training=True)
        m2._launch_backward(i2["x"], i2["t"], i2["length"], i2["xf_out"], saved, i2["x0"], want_dx=False)
args2 = types.SimpleNamespace(device=dev, diffusion_steps=1000, is_train=True, lr=2e-4, batch_size=16, num_epochs=1, log_every=50,
                              save_latest=500, save_every_e=5, is_continue=False, model_dir="/tmp", multi=True, label_path=None, cap_id=False)
tr2 = hig_amd.DDPMMulTrainer(args2, m2.train())
x0p, tp, lp = i2["x0"][:32].contiguous(), i2["t"][:16].contiguous(), i2["length"][:16].contiguous()
nz2 = torch.randn_like(x0p)
ls = []
for k in range(12):
    tr2.train_step_captured(x0p, tp, lp, i2["xf_proj"], i2["xf_out"], noise=nz2); ls.append("%.4f" % tr2.fused_state()["loss"].item())
print("f32 :", " ".join(ls))
m2b = hig_amd.MotionInteractionTransformer(input_feats=c2["F"], num_frames=196, latent_dim=c2["d"], ff_size=c2["ff"], num_layers=c2["L"],
                                           num_heads=c2["H"], text_latent_dim=c2["Lt"], storage="bf16")
m2b.load_state_dict(m2.state_dict())
m2b = m2b.to(dev)
tr2b = hig_amd.DDPMMulTrainer(args2, m2b.train())
ls = []
step = tr2b.train_step_fused if "capture16" in skip else tr2b.train_step_captured
nosync = "sync" not in skip
tj = []
for k in range(12):
    step(x0p, tp, lp, i2["xf_proj"], i2["xf_out"], noise=nz2)
    if k == 1 and "sync2" not in skip: torch.cuda.synchronize()
    if nosync: tj.append(tr2b.fused_state()["loss"].clone())
    else: ls.append("%.4f/%.2f" % (tr2b.fused_state()["loss"].item(), tr2b.fused_state()["gnorm"].item()))
print("bf16 (%s):" % ("back to back" if nosync else "host sync after every step"), " ".join(ls) if not nosync else " ".join("%.4f" % v.item() for v in tj))
