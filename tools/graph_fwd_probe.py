import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda", 0)
c = dict(bench.CFG)
m = bench.build_model(c, dev).eval()
i = bench.make_inputs(c, dev, 0)
def fwd():
    with torch.no_grad():
        return m(i["x"], i["t"], length=i["length"], xf_proj=i["xf_proj"], xf_out=i["xf_out"])
for _ in range(5): fwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): fwd()
torch.cuda.synchronize()
print("eager ms", (time.perf_counter() - t0) / 30 * 1e3)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fwd()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fwd()
for _ in range(5): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): g.replay()
torch.cuda.synchronize()
print("graph ms", (time.perf_counter() - t0) / 30 * 1e3)
