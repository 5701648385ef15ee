"""Diagnosis: captured bf16 training step, loss by step, with a device synchronisation after the second step (bench.py's timed():
two warm-up steps, synchronize, K steps back to back) or without any.  usage: sync_pattern_probe.py [single|pair]"""
import os, sys, types, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, hig_amd
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "single"
c = dict(bench.CFG)
if which == "pair": c.update(B=64, T=91, F=263)
i = bench.make_inputs(c, dev, 0)
args = types.SimpleNamespace(device=dev, diffusion_steps=1000, is_train=True, lr=2e-4, batch_size=c["B"], num_epochs=1, log_every=50,
                             save_latest=500, save_every_e=5, is_continue=False, model_dir="/tmp", multi=True, label_path=None, cap_id=False)
for sync_at in (None, 1):
    torch.manual_seed(0)
    if which == "pair":
        m = hig_amd.MotionInteractionTransformer(input_feats=c["F"], num_frames=196, latent_dim=c["d"], ff_size=c["ff"], num_layers=c["L"],
                                                 num_heads=c["H"], text_latent_dim=c["Lt"], storage=os.environ.get("ST", "bf16"))
        with torch.no_grad():
            for name, p in m.named_parameters():
                if name.startswith("out") or ".ffn.linear2." in name or ".out_layers.2." in name: p.copy_(torch.randn(p.shape) * 0.02)
        tr = hig_amd.DDPMMulTrainer(args, m.to(dev).train())
        x0, t, ln = i["x0"][:32].contiguous(), i["t"][:16].contiguous(), i["length"][:16].contiguous()
    else:
        m = bench.build_model(c, dev).train(); m.storage = os.environ.get("ST", "bf16")
        tr = hig_amd.DDPMTrainer(args, m)
        x0, t, ln = i["x0"], i["t"], i["length"]
    torch.manual_seed(1)
    nz = torch.randn_like(x0)
    tj = []
    for k in range(8):
        tr.train_step_captured(x0, t, ln, i["xf_proj"], i["xf_out"], noise=nz)
        tj.append(tr.fused_state()["loss"].clone())
        if sync_at is not None and k == sync_at: torch.cuda.synchronize()
    print(which, "sync after step 2" if sync_at is not None else "no sync", " ".join("%.4f" % v.item() for v in tj))
    del tr, m
