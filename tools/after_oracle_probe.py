"""Is the eager forward slower right after the CPU oracle ran on the host (OpenMP workers still spinning)?"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0); c = dict(bench.CFG)
m = bench.build_model(c, dev).eval(); i = bench.make_inputs(c, dev, 0)
m.cache_text_context = False
def fwd():
    with torch.no_grad():
        return m(i["x"], i["t"], length=i["length"], xf_proj=i["xf_proj"], xf_out=i["xf_out"])
def t(n=10):
    for _ in range(2): fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fwd()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for mode in ("bf16x3", "bf16", "bf16s", "bf16", "bf16"):
    bench.set_mode(m, mode)
    a = t()
    out = fwd()
    e = bench.oracle_slice_error(c, m, i, out)
    b = t()
    time.sleep(2.0)
    d = t()
    print("%-7s before oracle %.3f ms | right after the CPU oracle %.3f ms | 2 s later %.3f ms   (threads %d)" % (mode, a, b, d, torch.get_num_threads()))
