"""Probe: does running two half-batches on two HIP streams (tails of one overlapping heads of the other)
beat one full batch on one stream?  (tuning aid)"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0)
c = dict(bench.CFG)
m = bench.build_model(c, dev).eval()
i = bench.make_inputs(c, dev, 0)
m._pool.give = lambda *a, **k: None          # never recycle a workspace across streams in this probe
def fwd(sl):
    with torch.no_grad():
        return m(i["x"][sl], i["t"][sl], length=i["length"][sl], xf_proj=i["xf_proj"][sl], xf_out=i["xf_out"][sl])
full, h0, h1 = slice(0, 64), slice(0, 32), slice(32, 64)
m._textctx_cache = None
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("one stream, B=64:", timeit(lambda: fwd(full)))
print("one stream, 2 x B=32 back to back:", timeit(lambda: (fwd(h0), fwd(h1))))
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    m._textctx_cache = None
    with torch.cuda.stream(s0): fwd(h0)
    m._textctx_cache = None
    with torch.cuda.stream(s1): fwd(h1)
print("two streams, 2 x B=32:", timeit(two))
