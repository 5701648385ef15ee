"""Data-parallel training step: two ranks, ONE GPU, gloo backend -- the collective is the
same `dist.all_reduce(SUM)` of the flat gradient the RCCL run issues (tools/train.py:77-82 is what it replaces).
Pin: gradients after the exchange == mean over ranks of the per-rank gradients, and the parameters are identical on
all ranks after every step."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def test_two_rank_fused_training_step_matches_mean_gradient_step(dp_workers):
    import dp_worker
    outdir, codes, logs = dp_workers          # two ranks of tests/dp_worker.py, started at session start (conftest.py)
    world = 2
    assert codes == [0, 0], "\n".join(logs)[-4000:]
    got = [torch.load(os.path.join(outdir, "rank%d.pt" % r)) for r in range(world)]
    assert got[0]["step"] == got[1]["step"] == 2
    # replicas stay bit-identical: same averaged gradient, same clip coefficient, same Adam update on every rank --
    # although rank 1 STARTED from perturbed parameters (sync_replicas) and each rank saw different samples
    assert torch.equal(got[0]["flat"], got[1]["flat"])
    assert got[0]["gnorm"] == got[1]["gnorm"]
    assert got[0]["losses"] != got[1]["losses"]

    # single-process restatement: per-rank gradients one after the other, summed, then ONE clip+Adam with 1/world
    c, m, tr, gi, x0, noise = dp_worker.build_and_inputs()
    fp = m.flat_params()
    n = fp.core_numel
    for s in range(2):
        total = torch.zeros(n, device="cuda")
        for r in range(world):
            sh = lambda t: dp_worker.shard(t, r, world)
            tr._fused_fwd_bwd(sh(x0[s]), sh(gi["t"]), sh(gi["length"]), sh(gi["xf_proj"]), sh(gi["xf_out"]), sh(noise[s]))
            assert abs(tr.fused_state()["loss"].item() - got[r]["losses"][s]) <= 1e-6 * abs(got[r]["losses"][s])
            total += fp.grad[:n]
        fp.grad[:n].copy_(total)
        tr._fused_clip_adam(world)
    assert abs(tr.fused_state()["gnorm"].item() - got[0]["gnorm"]) <= 1e-6 * got[0]["gnorm"]
    ref = fp.flat[:n].cpu()
    # same kernels on the same numbers (a two-term sum commutes): expect bitwise equality; gate at fp32 rounding
    assert (ref - got[0]["flat"]).abs().max().item() <= 1e-7 * ref.abs().max().item()
