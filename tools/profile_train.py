#!/usr/bin/env python3
"""Runs a few fused training steps at BASELINE config 2 (for rocprofv3 --kernel-trace --stats)."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import hig_amd  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    c = dict(bench.CFG)
    m = bench.build_model(c, dev).train()
    inp = bench.make_inputs(c, dev, 0)
    args = types.SimpleNamespace(device=dev, diffusion_steps=1000, is_train=True, lr=2e-4, batch_size=c["B"],
                                 num_epochs=1, log_every=50, save_latest=500, save_every_e=5, is_continue=False,
                                 model_dir="/tmp")
    tr = hig_amd.DDPMTrainer(args, m)
    noise = torch.randn_like(inp["x0"])
    n = int(os.environ.get("STEPS", 6))
    for _ in range(n):
        tr.train_step_fused(inp["x0"], inp["t"], inp["length"], inp["xf_proj"], inp["xf_out"], noise=noise)
    torch.cuda.synchronize()
    print("loss", tr.fused_state()["loss"].item())
