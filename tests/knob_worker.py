"""Child process of tests/test_gpu_knobs.py: runs one small pass over every code path the library's environment switches touch
(the switches are read once per process) and prints a digest -- norms of the outputs -- as one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import hig_amd  # noqa: E402
from oracle import fill  # noqa: E402

DEV = "cuda"


def build(c, **kw):
    m = hig_amd.MotionTransformer(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"], ff_size=c["ff"],
                                  num_layers=c["L"], num_heads=c["H"], text_latent_dim=c["Lt"], **kw)
    m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
    return m.to(DEV)


def inputs(c):
    inp = fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"])
    return {k: v.to(DEV) for k, v in inp.items()}


def norm(t):
    return float(t.double().norm().item())


W = dict(fill.CASES["width"], L=2)                        # the config-2 model (d = 512, 8 heads of 64), two layers deep
big = dict(W, B=48, lengths=(196, 77, 196, 1, 120, 196, 50, 196) * 6, t=(0, 999, 500, 250, 7, 650, 313, 900) * 6)   # 9 408 rows
mid = dict(W, B=16, lengths=(196, 77, 196, 1, 120, 196, 50, 196) * 2, t=(0, 999, 500, 250, 7, 650, 313, 900) * 2)   # 3 136 rows
wide = dict(B=8, T=300, F=150, d=1024, H=8, L=2, ff=1024, N=77, Lt=256, num_frames=300,
            lengths=(300, 211, 300, 1, 150, 299, 64, 300), t=(5, 900, 0, 999, 313, 650, 77, 500))                  # head dim 128
out = {}
with torch.no_grad():
    gi = inputs(big)
    m = build(big).eval()
    out["f32_fwd_rows9408"] = norm(m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"]))
    m.cache_text_context = False      # the reference's per-call form: text side inside the call (forked / batched: HIG_TEXT_FORK, HIG_TEXT_BATCH)
    out["f32_fwd_per_call_rows9408"] = norm(m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"]))
    m.cache_text_context = True
    m.storage = "bf16"
    out["bf16_fwd_rows9408"] = norm(m(gi["x"], gi["t"], length=gi["length"], xf_proj=gi["xf_proj"], xf_out=gi["xf_out"]))
    del m
    gw = inputs(wide)
    mw = build(wide, storage="bf16").eval()
    out["bf16_fwd_d1024"] = norm(mw(gw["x"], gw["t"], length=gw["length"], xf_proj=gw["xf_proj"], xf_out=gw["xf_out"]))
    del mw
    small = dict(mid, B=2, lengths=(196, 77), t=(0, 999))
    gs = inputs(small)
    mn = build(small, no_eff=True).eval()
    out["f32_no_eff_fwd"] = norm(mn(gs["x"], gs["t"], length=gs["length"], xf_proj=gs["xf_proj"], xf_out=gs["xf_out"]))
    del mn
for storage in ("f32", "bf16"):
    gm = inputs(mid)
    mt = build(mid, storage=storage).train()
    x = gm["x"].clone().requires_grad_(True)
    o = mt(x, gm["t"], length=gm["length"], xf_proj=gm["xf_proj"], xf_out=gm["xf_out"])
    loss = (o ** 2).mean()
    loss.backward()
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in mt.parameters() if p.grad is not None)).item()
    out["%s_train_loss" % storage], out["%s_train_gradnorm" % storage], out["%s_train_dx" % storage] = float(loss.item()), gn, norm(x.grad)
    del mt
torch.cuda.synchronize()
print("DIGEST " + json.dumps(out))
