"""Prints a digest of every output of one forward + backward at config 2 and the two-person shape (to compare runs
that must be bitwise identical, e.g. HIG_BWD_OVERLAP=0 against 1)."""
import os, sys, zlib, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, hig_amd
dev = torch.device("cuda", 0)
def digest(ts):
    h = 0
    for t in ts:
        h = zlib.crc32(t.detach().cpu().contiguous().numpy().tobytes(), h)
    return "%08x" % h
for prec in ("f32", "bf16x3"):
    c = dict(bench.CFG)
    m = bench.build_model(c, dev).train(); m.precision = prec
    i = bench.make_inputs(c, dev, 0)
    for rep in range(3):
        out, saved = m._launch_forward(i["x"], i["t"], i["length"], i["xf_proj"], i["xf_out"], training=True)
        dx, dxp, dxo = m._launch_backward(i["x"], i["t"], i["length"], i["xf_out"], saved, i["x0"], want_dx=True)
        print(prec, rep, digest([out, dx, dxp, dxo, m.flat_params().grad]))
