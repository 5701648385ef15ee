"""bf16-storage training step at BASELINE config 2 next to the fp32 step (tuning aid): eager and hipGraph-captured, plus the
forward / forward+backward split.  CFG_B / CFG_T / CFG_d / CFG_L override the shape; STEPS the timed repeats."""
import os, sys, time, types, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, hig_amd
dev = torch.device("cuda", 0); c = dict(bench.CFG)
for k in ("B", "T", "d", "L"):
    if os.environ.get("CFG_" + k): c[k] = int(os.environ["CFG_" + k])
n = int(os.environ.get("STEPS", "10"))
i = bench.make_inputs(c, dev, 0)
args = types.SimpleNamespace(device=dev, diffusion_steps=1000, is_train=True, lr=2e-4, batch_size=c["B"], num_epochs=1,
                             log_every=50, save_latest=500, save_every_e=5, is_continue=False, model_dir="/tmp")
noise = torch.randn_like(i["x0"])
def t(fn, n=n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("B=%d T=%d d=%d L=%d" % (c["B"], c["T"], c["d"], c["L"]))
for storage in (os.environ.get("STORAGE", "f32,bf16").split(",")):
    m = bench.build_model(c, dev).train(); m.storage = storage
    tr = hig_amd.DDPMTrainer(args, m)
    if os.environ.get("ONLY_CAPTURED"):
        g = t(lambda: tr.train_step_captured(i["x0"], i["t"], i["length"], i["xf_proj"], i["xf_out"], noise=noise))
        print("storage %-5s captured %.3f ms" % (storage, g)); continue
    e = t(lambda: tr.train_step_fused(i["x0"], i["t"], i["length"], i["xf_proj"], i["xf_out"], noise=noise))
    loss = tr.fused_state()["loss"].item()
    if os.environ.get("NO_CAPTURE"):
        print("storage %-5s eager %.3f ms  (loss %.5f)" % (storage, e, loss)); continue
    g = t(lambda: tr.train_step_captured(i["x0"], i["t"], i["length"], i["xf_proj"], i["xf_out"], noise=noise))
    def fb():
        x = i["x0"].clone().requires_grad_(False)
        out = m(x, i["t"], length=i["length"], xf_proj=i["xf_proj"], xf_out=i["xf_out"])
        out.backward(noise)
    print("storage %-5s eager %.3f ms  captured %.3f ms  autograd fwd+bwd %.3f ms  (loss %.5f)" % (storage, e, g, t(fb), loss))
    del tr, m
