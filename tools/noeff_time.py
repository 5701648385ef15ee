import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda", 0)
for name, c in (("config 2", dict(bench.CFG)), ("config-5 shape", dict(bench.CFG, B=32, T=300, d=1024, L=12, H=8, ff=1024))):
    m = bench.build_model(c, dev, no_eff=True).eval(); i = bench.make_inputs(c, dev, 0)
    m.cache_text_context = False
    for mode in ("f32", "bf16s"):
        bench.set_mode(m, mode)
        def fwd():
            with torch.no_grad():
                return m(i["x"], i["t"], length=i["length"], xf_proj=i["xf_proj"], xf_out=i["xf_out"])
        for _ in range(3): fwd()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fwd()
        torch.cuda.synchronize(); print("no_eff %s %s: %.3f ms" % (name, mode, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
