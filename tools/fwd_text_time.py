"""Headline forward (text side recomputed per call) with the text side forked onto the library's third stream vs in front
(HIG_TEXT_FORK=0 in another process) -- prints ms per forward and a digest of the output."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0); c = dict(bench.CFG); c["B"] = int(os.environ.get("B", 64))
m = bench.build_model(c, dev).eval(); i = bench.make_inputs(c, dev, 0)
m.cache_text_context = os.environ.get("CACHE", "0") == "1"
def fwd():
    with torch.no_grad():
        return m(i["x"], i["t"], length=i["length"], xf_proj=i["xf_proj"], xf_out=i["xf_out"])
for _ in range(5): o = fwd()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): o = fwd()
torch.cuda.synchronize()
print("B=%d cache=%s HIG_TEXT_FORK=%s fwd ms %.3f  digest %.9e" % (c["B"], m.cache_text_context, os.environ.get("HIG_TEXT_FORK", "1"),
      (time.perf_counter() - t0) / 30 * 1e3, o.double().abs().sum().item()))
