"""Wall time of one forward + backward at BASELINE config 2 (tuning aid for env knobs)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0); c = dict(bench.CFG); m = bench.build_model(c, dev).train(); i = bench.make_inputs(c, dev, 0)
m.precision = os.environ.get("HIG_PREC", "f32")
def step():
    out, saved = m._launch_forward(i["x"], i["t"], i["length"], i["xf_proj"], i["xf_out"], training=True)
    m._launch_backward(i["x"], i["t"], i["length"], i["xf_out"], saved, i["x0"], want_dx=False)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("fwd+bwd ms %.3f" % ((time.perf_counter() - t0) / 10 * 1e3))
