import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import hig_amd
from oracle import fill
c = fill.CASES["config1"]
def build():
    m = hig_amd.MotionTransformer(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"], ff_size=c["ff"], num_layers=c["L"], num_heads=c["H"], text_latent_dim=c["Lt"])
    m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
    return m.to("cuda").eval()
gi = {k: v.to("cuda") for k, v in fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"]).items()}
def fwd(m):
    with torch.no_grad():
        return m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"])
m = build(); a = fwd(m); a2 = fwd(m)
f = build(); b = fwd(f); b2 = fwd(f)
print("m vs m", (a - a2).abs().max().item(), " f vs f", (b - b2).abs().max().item(), " m vs f", (a - b).abs().max().item(), " finite", torch.isfinite(a).all().item(), torch.isfinite(b).all().item())
with torch.no_grad():
    m.temporal_decoder_blocks[1].ca_block.key.weight.mul_(1.5)
c1 = fwd(m)
g = build()
with torch.no_grad():
    g.temporal_decoder_blocks[1].ca_block.key.weight.mul_(1.5)
c2 = fwd(g)
print("after mutation m vs fresh", (c1 - c2).abs().max().item(), "rel", ((c1 - c2).norm() / c2.norm()).item())
