"""Data-parallel pieces of the hot path: the ONE collective of the reference's training step and
its dataset sharding rule.

Reference: tools/train.py:53-90 wraps the model in (MM)DistributedDataParallel over a Gloo group
-- a bucketed all-reduce(sum)/world of every trainable gradient during backward -- and
datasets/dataloader.py:16-53 shards indices rank-strided with wrap-around padding.

MI355X design: one process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI).  All
core gradients already live in ONE flat fp32 buffer (models/transformer.py:_FlatParams), so the
exchange is a single all-reduce of ~324 MB per step; the 1/world scale is folded into the fused
clip+Adam kernel.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): one large message lets
RCCL use every link at once; 25 MB DDP buckets would be latency-bound per link.
"""
import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """all_reduce(SUM) of a flat gradient buffer; returns the world size (the caller divides).
    world == 1 (or no process group) is a no-op, so single-GPU runs need no rendezvous."""

    def __init__(self, group=None):
        self.group = group

    def __call__(self, flat_grad):
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        world = dist.get_world_size(self.group)
        if world > 1:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        return world


class ShardedSampler(torch.utils.data.Sampler):
    """Index sharding of the reference's DistributedSampler (datasets/dataloader.py:16-53):
    permutation seeded by `epoch` (the reference never calls set_epoch, so epoch stays 0),
    wrap-around padding to a multiple of world, then rank-strided subsampling."""

    def __init__(self, dataset_len, rank=0, world_size=1, shuffle=True, round_up=True):
        self.n, self.rank, self.world = int(dataset_len), int(rank), int(world_size)
        self.shuffle, self.round_up, self.epoch = shuffle, round_up, 0
        self.num_samples = -(-self.n // self.world)
        self.total_size = self.num_samples * self.world if round_up else self.n

    def set_epoch(self, epoch):
        self.epoch = epoch

    def indices(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if self.round_up:
            idx = (idx * int(self.total_size / len(idx) + 1))[:self.total_size]
        return idx[self.rank:self.total_size:self.world]

    def __iter__(self):
        return iter(self.indices())

    def __len__(self):
        return len(self.indices())


def broadcast_parameters(model, src=0, group=None):
    """Rank `src`'s parameters to every rank once at start (DDP does the same at wrap time)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in model.parameters():
            dist.broadcast(p.data, src=src, group=group)


def broadcast_flat(buf, src=0, group=None):
    """One broadcast of a flat device buffer (all parameters of the fused step, or its Adam moments)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(buf, src=src, group=group)
