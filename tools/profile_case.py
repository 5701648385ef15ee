#!/usr/bin/env python3
"""Runs one named workload a few times (for `rocprofv3 --kernel-trace --stats -- python3 tools/profile_case.py CASE`).
CASE: two_fwd | two_fwdbwd | text | fwd | fwdbwd | c5_fwd | c5_fwdbwd | c3_fwd (HIG_PREC selects the products)   (env STEPS = iterations, default 6)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import hig_amd  # noqa: E402

if __name__ == "__main__":
    case = sys.argv[1] if len(sys.argv) > 1 else "two_fwd"
    n = int(os.environ.get("STEPS", 6))
    dev = torch.device("cuda", 0)
    c = dict(bench.CFG)
    if case.startswith("two"):
        c.update(B=64, T=91, F=263)
        torch.manual_seed(0)
        m = hig_amd.MotionInteractionTransformer(input_feats=c["F"], num_frames=196, latent_dim=c["d"], ff_size=c["ff"],
                                                 num_layers=c["L"], num_heads=c["H"], text_latent_dim=c["Lt"])
        with torch.no_grad():
            for name, p in m.named_parameters():
                if name.startswith("out") or ".ffn.linear2." in name or ".out_layers.2." in name:
                    p.copy_(torch.randn(p.shape) * 0.02)
        m = m.to(dev)
    else:
        if case.startswith("c5"):          # BASELINE config 5 shape (hd = 128)
            c.update(B=32, T=300, d=1024, L=12, H=8, ff=1024)
        if case.startswith("c3"):          # BASELINE config 3 shape (sampling batch)
            c.update(B=32)
        m = bench.build_model(c, dev)
        m.precision = os.environ.get("HIG_PREC", "f32")
    i = bench.make_inputs(c, dev, 0)
    if case == "text":
        caps = ["a person walks towards another person and shakes hands number %d" % k for k in range(c["B"])]
        m.train()
        tok, feat = m._clip_features(caps, dev)
        for _ in range(n):
            xp, xo = m._text_head(tok, feat)
            (xp.sum() + xo.sum()).backward()
    elif case.endswith("fwdbwd"):
        for _ in range(n):
            out, saved = m._launch_forward(i["x"], i["t"], i["length"], i["xf_proj"], i["xf_out"], training=True)
            m._launch_backward(i["x"], i["t"], i["length"], i["xf_out"], saved, i["x0"], want_dx=False)
    else:
        with torch.no_grad():
            for _ in range(n):
                m(i["x"], i["t"], length=i["length"], xf_proj=i["xf_proj"], xf_out=i["xf_out"])
    torch.cuda.synchronize()
    print("done", case)
