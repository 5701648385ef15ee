"""bf16-storage forward of the two-person model (32 pairs x 91 tokens x 263 features), text side hoisted, eager and as a
replayed hipGraph.  usage: two_person16_time.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, hig_amd
dev = torch.device("cuda", 0)
c2 = dict(bench.CFG, B=64, T=91, F=263)
torch.manual_seed(0)
m2 = hig_amd.MotionInteractionTransformer(input_feats=c2["F"], num_frames=196, latent_dim=c2["d"], ff_size=c2["ff"],
                                          num_layers=c2["L"], num_heads=c2["H"], text_latent_dim=c2["Lt"])
with torch.no_grad():
    for name, p in m2.named_parameters():
        if name.startswith("out") or ".ffn.linear2." in name or ".out_layers.2." in name:
            p.copy_(torch.randn(p.shape) * 0.02)
m2 = m2.to(dev).eval(); m2.storage = "bf16"; m2.cache_text_context = True
i2 = bench.make_inputs(c2, dev, 0)
def fwd():
    with torch.no_grad():
        return m2(i2["x"], i2["t"], length=i2["length"], xf_proj=i2["xf_proj"], xf_out=i2["xf_out"])
for _ in range(5): fwd()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): fwd()
torch.cuda.synchronize(); print("two-person bf16 storage, eager: %.3f ms/forward" % ((time.perf_counter() - t0) / 50 * 1e3))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): fwd()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g): out = fwd()
for _ in range(5): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize(); print("   hipGraph replay: %.3f ms/forward" % ((time.perf_counter() - t0) / 200 * 1e3))
