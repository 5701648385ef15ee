"""Loss trajectory of the two-person PIT training step (bench.py's setup: 16 pairs x 91 tokens x 263 features, d = 512, L = 8) with
fp32 and bf16 storage, same data every step -- a diagnosis aid: the two must track each other.
usage: pit16_loss.py [steps] [captured=0|1]"""
import os, sys, types, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, hig_amd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
captured = len(sys.argv) > 2 and sys.argv[2] == "1"
dev = torch.device("cuda", 0)
c2 = dict(bench.CFG, B=64, T=91, F=263)
L = int(os.environ.get("CFG_L", c2["L"]))
torch.manual_seed(0)
def build(storage):
    torch.manual_seed(0)
    m = hig_amd.MotionInteractionTransformer(input_feats=c2["F"], num_frames=196, latent_dim=c2["d"], ff_size=c2["ff"], num_layers=L,
                                             num_heads=c2["H"], text_latent_dim=c2["Lt"], storage=storage)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.startswith("out") or ".ffn.linear2." in name or ".out_layers.2." in name:
                p.copy_(torch.randn(p.shape) * 0.02)
    return m.to(dev).train()
i2 = bench.make_inputs(c2, dev, 0)
args2 = types.SimpleNamespace(device=dev, diffusion_steps=1000, is_train=True, lr=2e-4, batch_size=16, num_epochs=1, log_every=50,
                              save_latest=500, save_every_e=5, is_continue=False, model_dir="/tmp", multi=True, label_path=None, cap_id=False)
x0p, tp, lp = i2["x0"][:32].contiguous(), i2["t"][:16].contiguous(), i2["length"][:16].contiguous()
nz2 = torch.randn_like(x0p)
for storage in ("f32", "bf16"):
    tr = hig_amd.DDPMMulTrainer(args2, build(storage))
    step = tr.train_step_captured if captured else tr.train_step_fused
    out = []
    for k in range(steps):
        step(x0p, tp, lp, i2["xf_proj"], i2["xf_out"], noise=nz2)
        out.append("%.4f/%.3f" % (tr.fused_state()["loss"].item(), tr.fused_state()["gnorm"].item()))
    print(storage, "captured" if captured else "eager", "loss/gnorm:", " ".join(out))
