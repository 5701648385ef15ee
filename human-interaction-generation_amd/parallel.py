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


class OverlappedGradAllReduce:
    """The same exchange, started per decoder layer WHILE the backward is still running (what the reference's DDP
    reducer does with its buckets, tools/train.py:77-82): hig_denoiser_bwd_hooked reports each layer as soon as its
    gradients are final, the layer's two flat ranges (its own parameters ~15 MB, its rows of the stacked stylization
    matrix ~25 MB at config 2) are all-reduced from a side stream, and only the global parameters (+ text head) are
    left for the end.  xGMI is point-to-point, so ~20-40 MB messages still run at link bandwidth; the exposed part of
    the exchange shrinks from the whole 324 MB to the last layer's bucket + the tail.

        ex = OverlappedGradAllReduce(); ex.begin(flat_grad, per_layer, tail)
        backward(layer_hook=ex.layer_done, comm_stream=ex.stream)
        world = ex.finish(extra_ranges)     # current stream now waits for every bucket
    """

    def __init__(self, group=None):
        self.group = group
        self.stream = None
        self.works = []

    @staticmethod
    def active(group=None):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def begin(self, flat_grad, per_layer, tail):
        if self.stream is None or self.stream.device != flat_grad.device:
            self.stream = torch.cuda.Stream(device=flat_grad.device)
        self.grad, self.per_layer, self.tail, self.works = flat_grad, per_layer, tail, []

    def _reduce(self, ranges):
        for a, b in ranges:
            if b > a:
                self.works.append(dist.all_reduce(self.grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def layer_done(self, layer):
        with torch.cuda.stream(self.stream):       # (the library already made this stream wait for the layer)
            self._reduce(self.per_layer[layer])

    def finish(self, extra=()):
        self.stream.wait_stream(torch.cuda.current_stream())     # the tail is final when the backward is
        with torch.cuda.stream(self.stream):
            self._reduce(list(self.tail) + list(extra))
        for w in self.works:
            w.wait()                                             # current stream waits for the exchange
        torch.cuda.current_stream().wait_stream(self.stream)
        self.works = []
        return dist.get_world_size(self.group)


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
