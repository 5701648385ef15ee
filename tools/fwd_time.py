"""Wall time of one inference forward (env B = batch, HIG_PREC = products, HIG_GEMM_TILE = forced tile)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0); c = dict(bench.CFG); c["B"] = int(os.environ.get("B", 64))
m = bench.build_model(c, dev).eval(); i = bench.make_inputs(c, dev, 0)
m.precision = os.environ.get("HIG_PREC", "f32")
def fwd():
    with torch.no_grad():
        return m(i["x"], i["t"], length=i["length"], xf_proj=i["xf_proj"], xf_out=i["xf_out"])
for _ in range(5): fwd()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): fwd()
torch.cuda.synchronize(); print("B=%d %s tile=%s fwd ms %.3f" % (c["B"], m.precision, os.environ.get("HIG_GEMM_TILE", "auto"), (time.perf_counter() - t0) / 30 * 1e3))
