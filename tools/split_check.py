"""Eager (batch halves on two streams) vs captured (one stream) fp32-storage forward: is the difference deterministic
rounding (tile schedule of the half batches) or a race?  usage: split_check.py [B] [precision]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda", 0)
c = dict(bench.CFG, B=B)
m = bench.build_model(c, dev).eval(); m.precision = prec
i = bench.make_inputs(c, dev, 0)
def fwd():
    with torch.no_grad():
        return m(i["x"], i["t"], length=i["length"], xf_proj=i["xf_proj"], xf_out=i["xf_out"])
outs = [fwd().clone() for _ in range(20)]
torch.cuda.synchronize()
print("B=%d %s: eager repeats identical: %s" % (B, prec, all(torch.equal(outs[0], o) for o in outs[1:])))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): fwd()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g): og = fwd()
g.replay(); torch.cuda.synchronize()
d = (og - outs[0]).double()
print("   captured vs eager: bitwise equal %s, rel-L2 %.3e, max abs %.3e (|out| max %.3e)" % (torch.equal(og, outs[0]), (d.norm() / outs[0].double().norm()).item(), d.abs().max().item(), outs[0].abs().max().item()))
