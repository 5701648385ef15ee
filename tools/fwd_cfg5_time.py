"""Forward time at the BASELINE config-5 shape (B = 32, T = 300, d = 1024, L = 12, H = 8, ff = 1024).
usage: python tools/fwd_cfg5_time.py [f32|bf16|bf16s]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16s"
dev = torch.device("cuda", 0)
c = dict(bench.CFG, B=32, T=300, d=1024, L=12, H=8, ff=1024)
m = bench.build_model(c, dev).eval(); i = bench.make_inputs(c, dev, 0)
bench.set_mode(m, mode)
m.cache_text_context = False
def fwd():
    with torch.no_grad():
        return m(i["x"], i["t"], length=i["length"], xf_proj=i["xf_proj"], xf_out=i["xf_out"])
for _ in range(3): fwd()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): fwd()
torch.cuda.synchronize(); print("config-5 shape, %s: %.3f ms/forward (text side included)" % (mode, (time.perf_counter() - t0) / 10 * 1e3))
