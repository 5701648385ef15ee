"""Eager fused vs hipGraph-captured training step at BASELINE config 2 (tuning aid); CFG_B / CFG_T / CFG_d / CFG_L
override the shape."""
import os, sys, time, types, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, hig_amd
dev = torch.device("cuda", 0); c = dict(bench.CFG)
for k in ("B", "T", "d", "L"):
    if os.environ.get("CFG_" + k): c[k] = int(os.environ["CFG_" + k])
m = bench.build_model(c, dev).train(); i = bench.make_inputs(c, dev, 0)
args = types.SimpleNamespace(device=dev, diffusion_steps=1000, is_train=True, lr=2e-4, batch_size=c["B"], num_epochs=1,
                             log_every=50, save_latest=500, save_every_e=5, is_continue=False, model_dir="/tmp")
tr = hig_amd.DDPMTrainer(args, m)
noise = torch.randn_like(i["x0"])
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("B=%d T=%d d=%d L=%d" % (c["B"], c["T"], c["d"], c["L"]))
print("eager fused  ms %.3f" % t(lambda: tr.train_step_fused(i["x0"], i["t"], i["length"], i["xf_proj"], i["xf_out"], noise=noise)))
print("captured     ms %.3f" % t(lambda: tr.train_step_captured(i["x0"], i["t"], i["length"], i["xf_proj"], i["xf_out"], noise=noise)))
