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


def test_two_rank_epoch_loop_shards_checkpoints_and_resumes(dp_train_workers):
    """`DDPMMulTrainer.train(dataset, rank, world=2)` (ddpm_trainer.py:220-266 + datasets/dataloader.py:16-53 +
    tools/train.py:53-90) as a whole: disjoint rank-strided shards of the SAME permutation in every epoch, replicas
    bit-identical after every phase, checkpoints written by rank 0 only, resume from rank 0's file, and the
    reference's own forward()/update() step averaging its gradients when handed a bare model in a 2-rank group."""
    from hig_amd.parallel import ShardedSampler
    outdir, codes, logs = dp_train_workers
    assert codes == [0, 0], "\n".join(logs)[-6000:]
    got = [torch.load(os.path.join(outdir, "train_rank%d.pt" % r)) for r in range(2)]
    n, bs = got[0]["n"], 2
    assert n == got[1]["n"] and n >= 5
    want = [ShardedSampler(n, r, 2).indices() for r in range(2)]
    per_epoch = (len(want[0]) // bs) * bs                      # drop_last
    assert sorted(want[0] + want[1])[:n] != [] and set(want[0] + want[1]) == set(range(n))   # together: every clip
    # (a resume re-enters the epoch index stored in the checkpoint, like the reference: `range(cur_epoch, num_epochs)`,
    # ddpm_trainer.py:226-241 -- phase 2 resumes at ep = 1 with num_epochs = 3 and so runs epochs 1 and 2)
    for tag, epochs in (("p1", 2), ("p2", 2), ("p3", 1)):
        for r in range(2):
            # the reference never calls set_epoch (dataloader.py:34-37): the same shard, in the same order, every epoch
            assert got[r][tag + ".seen"] == want[r][:per_epoch] * epochs, (tag, r)
        assert torch.equal(got[0][tag + ".params"], got[1][tag + ".params"]), tag      # replicas identical
        assert got[1][tag + ".saves"] == []                                            # only rank 0 writes
    steps = per_epoch // bs
    names = [s[0] for s in got[0]["p1.saves"]]
    assert names.count("latest.tar") >= 2 and "ckpt_e000.tar" in names and "ckpt_e001.tar" in names
    assert got[0]["p1.saves"][-2][1:] == (1, 2 * steps) or got[0]["p1.saves"][-1][1:] == (1, 2 * steps)
    ck = torch.load(os.path.join(outdir, "model", "latest.tar"), map_location="cpu")
    assert ck["ep"] == 2 and ck["total_it"] == 4 * steps                               # 2 epochs + the 2 resumed ones
    assert sorted(ck) == ["encoder", "ep", "opt_encoder", "total_it"]
    # the resumed epoch started from the checkpoint, not from the fresh (different) models: it moved on from phase 1
    assert not torch.equal(got[0]["p1.params"], got[0]["p2.params"])
    assert (got[0]["p1.params"] - got[0]["p2.params"]).abs().max().item() < 0.05


def test_bench_spawns_its_own_ranks_when_started_without_a_launcher(bench_spawn):
    """`python bench.py --gpus 2 ...` with no torchrun and no RANK / WORLD_SIZE in the environment (how a driver may start
    it): bench.py starts the two ranks itself before touching the GPU -- as the reference's launcher spawns its own ranks,
    tools/train.py:92-102 -- and relays rank 0's ONE JSON line.  On this 1-GPU box the ranks share the device over gloo."""
    import json
    outdir, codes, logs = bench_spawn
    assert codes == [0], "\n".join(logs)[-6000:]
    lines = [l for l in open(os.path.join(outdir, "stdout.txt"), errors="replace").read().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1
    assert r["value"] > 0 and r["ms_per_step"] > 0 and r["scaling"] == "weak"
    assert "single_gpu_reference" in r and r["config"]["parallelism"].startswith("dp2")
