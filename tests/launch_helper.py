"""Picklable rank bodies for the launcher tests (tests/test_cpu_host.py)."""
import os

import torch
import torch.distributed as dist


def allreduce_rank(rank, world, outdir, scale=1.0):
    t = torch.tensor([float(rank + 1) * scale])
    dist.all_reduce(t)
    with open(os.path.join(outdir, "rank%d.txt" % rank), "w") as f:
        f.write("%s %d %g" % (dist.get_backend(), world, t.item()))


def failing_rank(rank, world):
    raise ValueError("boom on rank %d" % rank)
