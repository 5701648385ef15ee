"""Forward / sampling-step time in the bf16-storage mode (and the fp32 reference points), with the per-kernel split
from torch.profiler-free HIP events.  usage: fwd16_time.py [B] [storage=bf16|f32] [precision]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
storage = sys.argv[2] if len(sys.argv) > 2 else "bf16"
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
dev = torch.device("cuda", 0)
c = dict(bench.CFG, B=B)
m = bench.build_model(c, dev).eval()
m.storage, m.precision = storage, prec
i = bench.make_inputs(c, dev, 0)
def fwd():
    with torch.no_grad():
        return m(i["x"], i["t"], length=i["length"], xf_proj=i["xf_proj"], xf_out=i["xf_out"])
for hoist in (True, False):
    m.cache_text_context = hoist
    for _ in range(5): fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fwd()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 50
    print("B=%d storage=%s prec=%s text %s: %.3f ms/forward (eager launches)" % (B, storage, prec, "hoisted" if hoist else "per call", el * 1e3))
m.cache_text_context = True
g = torch.cuda.CUDAGraph()
fwd(); torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): fwd()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g): out = fwd()
for _ in range(5): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize(); print("   hipGraph replay, text hoisted: %.3f ms/forward" % ((time.perf_counter() - t0) / 200 * 1e3))
