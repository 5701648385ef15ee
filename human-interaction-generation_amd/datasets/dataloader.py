"""`build_dataloader` of the reference (codes/datasets/dataloader.py:56-121) over `ShardedSampler`
(the same shuffle / round-up / rank-stride rule as its DistributedSampler, :16-53)."""
import random
from functools import partial

import numpy as np
from torch.utils.data import DataLoader

from ..parallel import ShardedSampler


def worker_init_fn(worker_id, num_workers, rank, seed):
    """dataloader.py:124-130: worker seed = num_workers * rank + worker_id + seed."""
    worker_seed = num_workers * rank + worker_id + seed
    np.random.seed(worker_seed)
    random.seed(worker_seed)


def build_dataloader(dataset, rank, world_size, samples_per_gpu, workers_per_gpu, num_gpus=1, dist=True,
                     shuffle=True, round_up=True, seed=None, persistent_workers=True, **kwargs):
    if dist:
        sampler = ShardedSampler(len(dataset), rank, world_size, shuffle=shuffle, round_up=round_up)
        shuffle = False
        batch_size, num_workers = samples_per_gpu, workers_per_gpu
    else:
        sampler = None
        batch_size, num_workers = num_gpus * samples_per_gpu, num_gpus * workers_per_gpu
    init_fn = partial(worker_init_fn, num_workers=num_workers, rank=rank, seed=seed) if seed is not None else None
    return DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers, pin_memory=False,
                      shuffle=shuffle, worker_init_fn=init_fn,
                      persistent_workers=persistent_workers and num_workers > 0, **kwargs)
