"""Probe: does running two half-batches on two HIP streams (tails of one overlapping heads of the other)
beat one full batch on one stream?  Text contexts are precomputed for every slice (tuning aid)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0)
c = dict(bench.CFG)
m = bench.build_model(c, dev).eval()
i = bench.make_inputs(c, dev, 0)
m._pool.give = lambda *a, **k: None          # never recycle a workspace across streams in this probe
sl = {"full": slice(0, 64), "h0": slice(0, 32), "h1": slice(32, 64)}
inp = {k: {n: v[s].clone() for n, v in i.items()} for k, s in sl.items()}
orig = m._text_context
ctx = {}
for k, d in inp.items():
    m._textctx_cache = None
    ctx[d["xf_out"].data_ptr()] = orig(m.dims(d["x"].shape[0], c["T"], c["N"]), d["xf_out"], False)
m._text_context = lambda dims, xf_out, training: ctx[xf_out.data_ptr()]
def fwd(k):
    d = inp[k]
    with torch.no_grad():
        return m(d["x"], d["t"], length=d["length"], xf_proj=d["xf_proj"], xf_out=d["xf_out"])
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("one stream, B=64: %.3f ms" % timeit(lambda: fwd("full")))
print("one stream, 2 x B=32 back to back: %.3f ms" % timeit(lambda: (fwd("h0"), fwd("h1"))))
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    with torch.cuda.stream(s0): fwd("h0")
    with torch.cuda.stream(s1): fwd("h1")
print("two streams, 2 x B=32: %.3f ms" % timeit(two))
