"""How much throughput is left on the table by running the step on ONE stream: N iterations of forward+backward
(or forward only) run here while a sibling process does the same on the same GPU.  If two concurrent processes take
clearly less than twice the solo time, independent work inside one step (weight gradients next to the data-gradient
chain) would fill the same gaps.  Usage: concurrency_probe.py [fwd|fwdbwd] [iters]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "fwdbwd"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0); c = dict(bench.CFG); m = bench.build_model(c, dev).train(); i = bench.make_inputs(c, dev, 0)
def step():
    out, saved = m._launch_forward(i["x"], i["t"], i["length"], i["xf_proj"], i["xf_out"], training=mode == "fwdbwd")
    if mode == "fwdbwd":
        m._launch_backward(i["x"], i["t"], i["length"], i["xf_out"], saved, i["x0"], want_dx=False)
for _ in range(3): step()
torch.cuda.synchronize()
# crude rendezvous with the sibling: both start timing at the next multiple of 5 s of wall-clock
t_go = (int(time.time()) // 5 + 1) * 5
while time.time() < t_go: pass
t0 = time.perf_counter()
for _ in range(iters): step()
torch.cuda.synchronize(); print("%s: %d iterations, %.3f ms each" % (mode, iters, (time.perf_counter() - t0) / iters * 1e3))
