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
import os
import socket

import torch
import torch.distributed as dist


def exchange_forced():
    """HIG_FORCE_EXCHANGE=1: issue the gradient all-reduces even in a group of ONE rank (where they are the identity).
    Lets a single GPU exercise the real RCCL path -- the collectives on the side stream, their ordering against the
    backward's two streams, the graph-A / all-reduce / graph-B split of the captured step -- without a second device
    (tests/test_gpu_rccl.py)."""
    return os.environ.get("HIG_FORCE_EXCHANGE", "0") == "1"


def exchange_active(group=None):
    """True when the training step has a gradient exchange to run: more than one rank, or one rank with the switch."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or exchange_forced()


class FlatGradAllReduce:
    """all_reduce(SUM) of a flat gradient buffer; returns the world size (the caller divides).
    world == 1 (or no process group) is a no-op, so single-GPU runs need no rendezvous."""

    def __init__(self, group=None):
        self.group = group

    def __call__(self, flat_grad):
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        world = dist.get_world_size(self.group)
        if world > 1 or exchange_forced():
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        return world


class OverlappedGradAllReduce:
    """The same exchange, started per decoder layer WHILE the backward is still running (what the reference's DDP
    reducer does with its buckets, tools/train.py:77-82): hig_denoiser_bwd_hooked / hig_denoiser_bwd_bf16_hooked report
    each layer as soon as its gradients are final, the layer's two flat ranges (its own parameters ~15 MB, its rows of
    the stacked stylization matrix ~25 MB at config 2) are all-reduced from a side stream, and only the global
    parameters (+ text head) are left for the end.  xGMI is point-to-point, so ~20-40 MB messages still run at link
    bandwidth; the exposed part of the exchange shrinks from the whole 324 MB to the last layer's bucket + the tail.

        ex = OverlappedGradAllReduce(); ex.begin(flat_grad, per_layer, tail)
        backward(layer_hook=ex.layer_done, comm_stream=ex.stream)
        world = ex.finish(extra_ranges)     # current stream now waits for every bucket

    The whole sequence may run under stream capture (DDPMTrainer.train_step_captured): the side stream joins the
    capture through the library's "layer done" event, RCCL's kernels become graph nodes, finish() joins everything back,
    and a replay costs the host nothing per collective.

    wire="bf16": each bucket travels as bf16 (half the bytes): cast into a staging buffer, all-reduce, cast back -- one
    extra rounding of every gradient element before the sum over ranks.  Opt-in (the reference's DDP exchanges fp32);
    tests/test_gpu_rccl.py pins what it does to the clip norm."""

    def __init__(self, group=None, wire="f32"):
        assert wire in ("f32", "bf16")
        self.group = group
        self.wire = wire
        self.stream = None
        self.stage = None

    @staticmethod
    def active(group=None):
        return exchange_active(group)

    def begin(self, flat_grad, per_layer, tail):
        if self.stream is None or self.stream.device != flat_grad.device:
            self.stream = torch.cuda.Stream(device=flat_grad.device)
        self.grad, self.per_layer, self.tail = flat_grad, per_layer, tail
        if self.wire == "bf16" and (self.stage is None or self.stage.numel() != flat_grad.numel()
                                    or self.stage.device != flat_grad.device):
            self.stage = torch.empty(flat_grad.numel(), device=flat_grad.device, dtype=torch.bfloat16)

    def _reduce(self, ranges):
        """Called with self.stream current.  Synchronous-style calls on purpose (no async_op, no Work handle kept): on the RCCL
        backend that only makes self.stream wait for RCCL's internal stream -- the order the collectives run in anyway -- and no
        Work object (with events recorded under a stream capture) outlives the call, so nothing the process group's watchdog
        thread could query is left behind when the step is being captured."""
        for a, b in ranges:
            if b <= a:
                continue
            if self.wire == "f32":
                dist.all_reduce(self.grad[a:b], op=dist.ReduceOp.SUM, group=self.group)
                continue
            from . import _lib
            sp = _lib.stream_ptr()
            _lib.check(_lib.lib().hig_cast_bf16(_lib.ptr(self.grad[a:b]), _lib.ptr(self.stage[a:b]), b - a, sp))
            dist.all_reduce(self.stage[a:b], op=dist.ReduceOp.SUM, group=self.group)      # (self.stream waits for RCCL's)
            _lib.check(_lib.lib().hig_cast_f32(_lib.ptr(self.stage[a:b]), _lib.ptr(self.grad[a:b]), b - a, sp))

    def layer_done(self, layer):
        with torch.cuda.stream(self.stream):       # (the library already made this stream wait for the layer)
            self._reduce(self.per_layer[layer])

    def finish(self, extra=()):
        self.stream.wait_stream(torch.cuda.current_stream())     # the tail is final when the backward is
        with torch.cuda.stream(self.stream):
            self._reduce(list(self.tail) + list(extra))
        torch.cuda.current_stream().wait_stream(self.stream)     # current stream waits for the exchange
        return dist.get_world_size(self.group)


def drain_pending_collectives(group=None):
    """Before a stream capture that contains collectives: every collective issued so far has finished on the GPU AND has been
    retired by the process group's watchdog thread (it polls the completion events of in-flight work; a capture must not start
    while it still holds any).  Deterministic replacement of the fixed sleep of round 5: ProcessGroupNCCL.waitForPendingWorks
    returns when the watchdog's list is empty."""
    torch.cuda.synchronize()
    if not (dist.is_available() and dist.is_initialized()):
        return
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    wait = getattr(pg, "_wait_for_pending_works", None)
    if wait is not None:
        wait()
    else:                                      # (older torch: the watchdog polls every 100 ms)
        import time
        time.sleep(0.3)


def all_ranks_agree(ok, device, group=None):
    """True only if `ok` holds on EVERY rank (one eager MIN all-reduce; a group of one rank answers for itself).  A decision
    that changes the sequence of collectives a rank will issue -- such as falling back from the captured exchange to the split
    form -- has to be taken by all ranks alike, or their collective sequences stop matching."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32,
                        device=device if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item())


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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch_entry(local_rank, fn, world_size, backend, args, kwargs, port, set_device):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the hosts only support dmabuf IPC
    run_distributed(fn, local_rank, world_size, backend, *args, set_device=set_device, **kwargs)


def run_distributed(fn, rank, world_size, backend=None, *args, set_device=True, **kwargs):
    """One rank of a data-parallel job: `setup` + `ddp_train` of the reference launcher (tools/train.py:53-90) around
    `fn(rank, world_size, *args, **kwargs)`.  Rendezvous from MASTER_ADDR / MASTER_PORT (torchrun's variables work
    unchanged); backend "nccl" (= RCCL over xGMI) when a GPU is present, else "gloo" (the reference's choice,
    tools/train.py:55) -- HIG_DIST_BACKEND overrides.  One process per GPU: the device is LOCAL_RANK when the launcher
    set it (torchrun), else rank % device_count.  Returns what `fn` returns.
    Barrier before `fn`, like the reference after creating its output directories; the group is destroyed afterwards,
    also when `fn` raises."""
    have_gpu = torch.cuda.device_count() > 0
    backend = backend or os.environ.get("HIG_DIST_BACKEND") or ("nccl" if have_gpu else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {}
    if have_gpu and set_device:
        local = int(os.environ["LOCAL_RANK"]) if "LOCAL_RANK" in os.environ else rank
        dev = torch.device("cuda", local % torch.cuda.device_count())
        torch.cuda.set_device(dev)
        if backend == "nccl":
            kw["device_id"] = dev
    # (RCCL collectives are captured into the training step's hipGraph: do not let the process group recycle events that were last
    # recorded under capture -- its watchdog thread may still poll a collective that holds one)
    os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")
    dist.init_process_group(backend, rank=rank, world_size=world_size, **kw)
    try:
        if world_size > 1:
            dist.barrier()
        return fn(rank, world_size, *args, **kwargs)
    finally:
        dist.destroy_process_group()


def launch(fn, world_size=None, backend=None, args=(), kwargs=None, port=None, set_device=True):
    """`main` of the reference launcher (tools/train.py:92-102): one process per GPU of this node running
    `fn(rank, world_size, *args)` inside an initialised process group.  Under torchrun (RANK / WORLD_SIZE in the
    environment) the processes already exist: this one just joins as its rank.  Otherwise `world_size` processes
    (default: HIG_WORLD_SIZE, else one per visible GPU, at least one) are spawned, rendezvous on 127.0.0.1 at a free
    port.  The parent launches no kernel and creates no HIP context of its own; counting the devices
    (`torch.cuda.device_count()`, only when neither `world_size` nor HIG_WORLD_SIZE is given) is the one query it makes.
    Return value: `fn`'s result when this process is itself a rank (torchrun), None after spawning (the children's
    results stay in the children, as with the reference's `mp.spawn`)."""
    kwargs = kwargs or {}
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        return run_distributed(fn, int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), backend, *args,
                               set_device=set_device, **kwargs)
    world_size = world_size or int(os.environ.get("HIG_WORLD_SIZE", "0")) or max(1, torch.cuda.device_count())
    port = port or _free_port()
    import torch.multiprocessing as mp
    mp.spawn(_launch_entry, args=(fn, world_size, backend, args, kwargs, port, set_device), nprocs=world_size, join=True)
