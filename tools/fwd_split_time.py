"""Experiment: one inference forward of B samples as two half batches on two streams (two model copies, so no scratch is
shared) against the single-stream forward.  usage: python tools/fwd_split_time.py [B] [storage]"""
import copy, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda", 0); c = dict(bench.CFG); c["B"] = B
m = bench.build_model(c, dev).eval(); i = bench.make_inputs(c, dev, 0)
bench.set_mode(m, mode)
m.cache_text_context = os.environ.get("HOIST", "0") == "1"
m2 = copy.deepcopy(m)
h = B // 2
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def full():
    with torch.no_grad():
        return m(i["x"], i["t"], length=i["length"], xf_proj=i["xf_proj"], xf_out=i["xf_out"])


xa, xb = i["x"][:h].contiguous(), i["x"][h:].contiguous()
oa, ob = i["xf_out"][:h].contiguous(), i["xf_out"][h:].contiguous()


def split():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.no_grad():
        with torch.cuda.stream(s1):
            a = m(xa, i["t"][:h], length=i["length"][:h], xf_proj=i["xf_proj"][:h], xf_out=oa)
        with torch.cuda.stream(s2):
            b = m2(xb, i["t"][h:], length=i["length"][h:], xf_proj=i["xf_proj"][h:], xf_out=ob)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b


ref = full(); a, b = split(); torch.cuda.synchronize()
print("max |split - full| = %.3e (|full| max %.3e)" % ((torch.cat([a, b]) - ref).abs().max().item(), ref.abs().max().item()))
for name, fn in (("full", full), ("split", split), ("full", full), ("split", split)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): fn()
    torch.cuda.synchronize()
    print("B=%d %s hoisted=%s  %-6s %.3f ms" % (B, mode, m.cache_text_context, name, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
