"""MI355X-native `DDPMTrainer`.

Public surface = what the reference's tools call on its trainer (codes/trainers/ddpm_trainer.py:29-266:
`forward` / `backward_G` / `update`, `generate` / `generate_batch`, `save` / `load`, `train`, the attribute
names and the checkpoint keys); the bodies are this project's own and are organised around ONE training step:

  * `train_step_fused(...)`   explicit launches: q_sample -> [text head] -> denoiser forward -> masked MSE ->
                              denoiser backward -> ONE flat-buffer all-reduce (RCCL) -> fused clip(0.5)+Adam;
  * `train_step_captured(...)` the same step replayed as hipGraph(s);
  * `forward(batch)` + `update()`  the reference's autograd / torch-Adam sequence, kept so driver code written
                              against the reference runs unchanged (and as the parity anchor of the fused step).

`opt.fused_step` makes `train()` take the fused step; its Adam moments then live in flat device buffers and are
written to / restored from checkpoints in torch-Adam's own `state_dict` layout, so a checkpoint resumes under either
step (and under the reference's trainer).
"""
import time
from collections import OrderedDict
from os.path import join as pjoin

import torch
import torch.distributed as dist
import torch.optim as optim
from torch.nn.utils import clip_grad_norm_

from .. import _lib
from ..models.gaussian_diffusion import (GaussianDiffusion, LossType, ModelMeanType, ModelVarType,
                                         create_named_schedule_sampler, get_named_beta_schedule)
from ..parallel import (FlatGradAllReduce, OverlappedGradAllReduce, ShardedSampler, broadcast_flat, broadcast_parameters,
                        exchange_active)

GRAD_CLIP = 0.5                    # ddpm_trainer.py:61
ADAM_BETAS, ADAM_EPS = (0.9, 0.999), 1e-8   # torch.optim.Adam defaults, ddpm_trainer.py:222


def _core(encoder):
    """The bare model under a DDP-style wrapper (the reference reaches for `.module` first, :116-119)."""
    return getattr(encoder, "module", encoder)


def _dist_world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


class _RunningMean:
    """Sums per-key values (host floats or device scalars -- the latter without any host sync) and hands back their
    means when the ITERATION COUNT is a multiple of `period` -- the reference prints at `it % log_every == 0`
    (ddpm_trainer.py:247-256), which after a resume from a checkpoint is not every `period` additions.  Only a rank
    that prints (`read=True`) reads the device scalars back; the others just drop the sums."""

    def __init__(self, period):
        self.period, self.n, self.acc = int(period), 0, OrderedDict()

    def add(self, values, it=None, read=True):
        for k, v in values.items():
            self.acc[k] = self.acc[k] + v if k in self.acc else (v.clone() if torch.is_tensor(v) else v)
        self.n += 1
        due = (it % self.period == 0) if it is not None else (self.n >= self.period)
        if not due:
            return None
        div = self.period if it is not None else self.n      # (the reference divides by log_every whatever it summed)
        out = OrderedDict((k, float(v) / div) for k, v in self.acc.items()) if read else None
        self.n, self.acc = 0, OrderedDict()
        return out


class DDPMTrainer(object):

    def __init__(self, args, encoder):
        self.opt, self.device, self.encoder = args, args.device, encoder
        self.multi = False
        self.diffusion_steps = args.diffusion_steps
        self.sampler_name = 'uniform'
        # the one diffusion every reference tool builds: linear betas, eps-prediction, fixed small variance, MSE
        self.diffusion = GaussianDiffusion(betas=get_named_beta_schedule('linear', self.diffusion_steps),
                                           model_mean_type=ModelMeanType.EPSILON,
                                           model_var_type=ModelVarType.FIXED_SMALL, loss_type=LossType.MSE)
        self.sampler = create_named_schedule_sampler(self.sampler_name, self.diffusion)
        if args.is_train:
            self.mse_criterion = torch.nn.MSELoss(reduction='none')
        self._fused = None
        self.to(self.device)

    # ---- small helpers the reference exposes as static methods ----------------------------------------------
    @staticmethod
    def zero_grad(opt_list):
        any(o.zero_grad() for o in opt_list)

    @staticmethod
    def clip_norm(network_list):
        any(clip_grad_norm_(net.parameters(), GRAD_CLIP) is None for net in network_list)

    @staticmethod
    def step(opt_list):
        any(o.step() for o in opt_list)

    def to(self, device):
        if hasattr(self, "mse_criterion"):
            self.mse_criterion.to(device)
        self.encoder = self.encoder.to(device)

    def train_mode(self):
        self.encoder.train()

    def eval_mode(self):
        self.encoder.eval()

    # ---- batch staging shared by both step flavours ---------------------------------------------------------
    def _stage_batch(self, batch_data):
        """(captions, motions (B,T,F), m_lens) -> captions, x_start on the device (fp32), lengths clamped to T."""
        caption, motions, m_lens = batch_data
        x_start = motions.detach().to(self.device).float()
        T = x_start.shape[1]
        cur_len = torch.tensor([min(T, int(n)) for n in m_lens], dtype=torch.int64).to(self.device)
        return caption, x_start, cur_len

    # ---- the reference's step: forward() then update() --------------------------------------------------------
    def forward(self, batch_data, eval_mode=False):
        """Noises the batch at sampled timesteps and runs the denoiser (ddpm_trainer.py:97-119); leaves
        `real_noise`, `fake_noise`, `src_mask` for `backward_G`."""
        caption, x_start, cur_len = self._stage_batch(batch_data)
        self.caption, self.motions = caption, x_start
        t, _unit_weights = self.sampler.sample(x_start.shape[0], x_start.device)
        terms = self.diffusion.training_losses(model=self.encoder, x_start=x_start, t=t,
                                               model_kwargs=dict(text=caption, length=cur_len))
        self.real_noise, self.fake_noise = terms['target'], terms['pred']
        self.src_mask = _core(self.encoder).generate_src_mask(x_start.shape[1], cur_len).to(x_start.device)

    def backward_G(self):
        """Masked mean over valid frames of the per-frame feature-mean squared error (ddpm_trainer.py:172-178)."""
        per_frame = self.mse_criterion(self.fake_noise, self.real_noise).mean(dim=-1)
        self.loss_mot_rec = (per_frame * self.src_mask).sum() / self.src_mask.sum()
        return OrderedDict(loss_mot_rec=self.loss_mot_rec.item())

    def update(self):
        """zero_grad -> loss -> backward -> clip_grad_norm_(0.5) -> Adam (ddpm_trainer.py:180-187)."""
        self.zero_grad([self.opt_encoder])
        logs = self.backward_G()
        self.loss_mot_rec.backward()
        self._exchange_param_grads()
        self.clip_norm([self.encoder])
        self.step([self.opt_encoder])
        _core(self.encoder).params_changed()
        return logs

    def _exchange_param_grads(self):
        """The reference ALWAYS trains through a DDP wrapper when world > 1 (tools/train.py:77-82), whose reducer
        averages the gradients during backward().  When this trainer is handed a bare model in a multi-rank group,
        the same averaging happens here (one all-reduce of the flat gradient buffer + one per parameter outside it)
        instead of silently stepping every rank on its own gradient.  A wrapped model (`.module`) already did it."""
        world = _dist_world()
        if world == 1 or hasattr(self.encoder, "module"):
            return
        # (under autograd the parameter gradients are tensors of their own -- _DenoiserFn.backward hands out copies of
        # the flat buffer's views -- so they are packed into one message here, like one DDP bucket).  The message has a
        # FIXED layout -- every trainable parameter in `parameters()` order, zeros where a rank produced no gradient --
        # so that all ranks pair the same elements (DDP's buckets are laid out once, too); the buffer and its views are
        # cached across steps.
        params = [p for p in self.encoder.parameters() if p.requires_grad]
        if not params:
            return
        key = tuple((id(p), p.numel()) for p in params)
        cache = getattr(self, "_exchange_cache", None)
        if cache is None or cache[0] != key or cache[1].device != params[0].device:
            flat = torch.zeros(sum(p.numel() for p in params), device=params[0].device, dtype=torch.float32)
            views, o = [], 0
            for p in params:
                views.append(flat[o:o + p.numel()].view(p.shape))
                o += p.numel()
            cache = self._exchange_cache = (key, flat, views)
        _, flat, views = cache
        for p, v in zip(params, views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / world)
        for p, v in zip(params, views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)

    # ---- sampling ----------------------------------------------------------------------------------------------
    def generate_batch(self, caption, m_lens, dim_pose):
        """One chunk of captions -> (B, T, dim_pose) samples: text encoded once, then the 1000-step loop (captured
        as a hipGraph by GaussianDiffusion); T = longest requested length, capped at num_frames (:121-150)."""
        core = _core(self.encoder)
        xf_proj, xf_out = core.encode_text(caption, self.device)
        T = min(int(m_lens.max()), core.num_frames)
        return self.diffusion.p_sample_loop(self.encoder, (len(caption), T, dim_pose), clip_denoised=False,
                                            progress=True,
                                            model_kwargs=dict(xf_proj=xf_proj, xf_out=xf_out, length=m_lens))

    def generate(self, caption, m_lens, dim_pose, batch_size=1024):
        """All captions in chunks of `batch_size` -> python list of (T_chunk, dim_pose) tensors (:152-170)."""
        self.encoder.eval()
        samples = []
        for lo in range(0, len(caption), batch_size):
            hi = min(len(caption), lo + batch_size)
            samples.extend(self.generate_batch(caption[lo:hi], m_lens[lo:hi], dim_pose).unbind(0))
        return samples

    # ---- checkpoints (keys: opt_encoder / ep / total_it / encoder, ddpm_trainer.py:200-218) ---------------------
    def _torch_optimizer(self):
        if not hasattr(self, "opt_encoder"):
            self.opt_encoder = optim.Adam(self.encoder.parameters(), lr=self.opt.lr)
        return self.opt_encoder

    def _fused_param_slots(self):
        """[(index in encoder.parameters(), flat offset, parameter)] for everything the fused Adam has stepped."""
        fp = _core(self.encoder).flat_params()
        index = {id(p): i for i, p in enumerate(self.encoder.parameters())}
        covered = self._fused["covered"] if self._fused else 0
        slots = zip(fp.params + fp.text_params, fp.offsets + fp.text_offsets)
        return [(index[id(p)], off, p) for p, off in slots if off < covered and id(p) in index]

    def _optimizer_state_dict(self):
        """torch.optim.Adam.state_dict() of the optimizer that is really in use: when the fused step has run, its
        flat moments / step counter are laid out as Adam's per-parameter `exp_avg` / `exp_avg_sq` / `step`."""
        sd = self._torch_optimizer().state_dict()
        st = self._fused
        if st is None or int(st["step"].item()) == 0:
            return sd
        step = torch.tensor(float(st["step"].item()))
        for idx, off, p in self._fused_param_slots():
            n = p.numel()
            sd['state'][idx] = {'step': step.clone(), 'exp_avg': st["m"][off:off + n].view(p.shape).clone(),
                                'exp_avg_sq': st["v"][off:off + n].view(p.shape).clone()}
        return sd

    def _adopt_optimizer_state(self, sd):
        """The inverse: Adam moments of a checkpoint into the fused step's flat buffers."""
        st = self.fused_state()
        fp = _core(self.encoder).flat_params()
        index = {id(p): i for i, p in enumerate(self.encoder.parameters())}
        steps, covered = [], 0
        with torch.no_grad():
            for p, off in zip(fp.params + fp.text_params, fp.offsets + fp.text_offsets):
                ent = sd['state'].get(index.get(id(p)))
                if ent is None:
                    continue
                n = p.numel()
                st["m"][off:off + n].copy_(ent['exp_avg'].reshape(-1))
                st["v"][off:off + n].copy_(ent['exp_avg_sq'].reshape(-1))
                steps.append(int(float(ent['step'])))
                covered = max(covered, off + n)
            st["step"].fill_(max(steps) if steps else 0)
        st["covered"] = max(st["covered"], covered)

    def save(self, file_name, ep, total_it):
        torch.save({'opt_encoder': self._optimizer_state_dict(), 'ep': ep, 'total_it': total_it,
                    'encoder': _core(self.encoder).state_dict()}, file_name)

    def load(self, model_dir):
        ck = torch.load(model_dir, map_location=self.device)
        core = _core(self.encoder)
        core.load_state_dict(ck['encoder'], strict=True)
        core.params_changed()
        if self.opt.is_train:
            self._torch_optimizer().load_state_dict(ck['opt_encoder'])
            if getattr(self.opt, "fused_step", False) or self._fused is not None:
                self._adopt_optimizer_state(ck['opt_encoder'])
        return ck['ep'], ck.get('total_it', 0)

    # ---- the epoch loop (ddpm_trainer.py:220-266: logging and checkpoint cadence) ----------------------------------
    def _loader(self, dataset, rank, world_size):
        sampler = ShardedSampler(len(dataset), rank, world_size, shuffle=True)
        return torch.utils.data.DataLoader(dataset, batch_size=self.opt.batch_size, sampler=sampler, drop_last=True,
                                           shuffle=False, num_workers=getattr(self.opt, "num_workers", 4))

    def sync_replicas(self):
        """Rank 0's trainable parameters (and fused Adam state) to every rank.  The fused step exchanges gradients
        itself instead of wrapping the model in DDP, so the construction-time broadcast DDP would do
        (tools/train.py:77-82) happens here.  Direct `train_step_fused` users with > 1 rank call this once first."""
        if _dist_world() == 1:
            return
        core = _core(self.encoder)
        fp = core.flat_params()
        broadcast_flat(fp.flat)
        in_flat = {id(p) for p in fp.params + fp.text_params}
        for p in core.parameters():
            if p.requires_grad and id(p) not in in_flat:
                dist.broadcast(p.data, src=0)
        if self._fused is not None:
            for k in ("m", "v"):
                broadcast_flat(self._fused[k])
            stp = self._fused["step"].to(torch.int64)
            dist.broadcast(stp, src=0)
            self._fused["step"].copy_(stp.to(torch.int32))
        core.params_changed()

    def _one_step(self, batch_data, fused):
        if fused:   # the loss stays on the device; `_RunningMean` only reads it back when it prints
            return OrderedDict(loss_mot_rec=self.train_fused_batch(
                batch_data, captured=getattr(self.opt, "fused_graph", False)))
        self.forward(batch_data)
        return self.update()

    def train(self, train_dataset, rank, world_size):
        self.to(self.device)
        self.opt_encoder = optim.Adam(self.encoder.parameters(), lr=self.opt.lr)
        latest = pjoin(self.opt.model_dir, 'latest.tar')
        first_epoch, it = self.load(latest) if self.opt.is_continue else (0, 0)
        fused = bool(getattr(self.opt, "fused_step", False))
        if world_size > 1 and _dist_world() != world_size:
            raise RuntimeError("train(world_size=%d) needs an initialised process group of that size (found %d rank(s)): "
                               "start the ranks through hig_amd.parallel.launch / torchrun" % (world_size, _dist_world()))
        if fused:
            self.sync_replicas()
        elif world_size > 1 and not hasattr(self.encoder, "module"):
            broadcast_parameters(self.encoder)      # what DDP's constructor does; the step averages the gradients itself
            _core(self.encoder).params_changed()
        loader = self._loader(train_dataset, rank, world_size)
        meter, t_start = _RunningMean(self.opt.log_every), time.time()
        for epoch in range(first_epoch, self.opt.num_epochs):
            self.train_mode()
            for i, batch_data in enumerate(loader):
                it += 1
                means = meter.add(self._one_step(batch_data, fused), it=it, read=(rank == 0))
                if means is not None and rank == 0:
                    print('epoch: %3d niter: %6d inner_iter: %4d %.0fs %s' % (
                        epoch, it, i, time.time() - t_start, ' '.join('%s: %.4f' % kv for kv in means.items())))
                if it % self.opt.save_latest == 0 and rank == 0:
                    self.save(latest, epoch, it)
            if rank == 0:
                self.save(latest, epoch, it)
                if epoch % self.opt.save_every_e == 0:
                    self.save(pjoin(self.opt.model_dir, 'ckpt_e%03d.tar' % epoch), epoch, total_it=it)

    # ------------------------------------------------------------------------------------------
    # MI355X fused step
    # ------------------------------------------------------------------------------------------
    def fused_state(self):
        """Flat Adam moments, device scalars (loss, grad norm, step, lr) and scratch of the fused step (lazy).
        Tied to the model's flat parameter buffer: when `.to()` re-homes the parameters the moments move along and
        every captured graph (which holds raw pointers into the old buffers) is dropped."""
        core = _core(self.encoder)
        fp = core.flat_params()
        fp.ensure_grad()
        ptrs = (fp.flat.data_ptr(), fp.grad.data_ptr())
        st = self._fused
        if st is not None and st["ptrs"] != ptrs:
            dev = fp.flat.device
            if st["m"].numel() == fp.flat.numel():
                for k in ("m", "v", "scratch", "mse_scratch", "loss", "gnorm", "step", "lr"):
                    st[k] = st[k].to(dev)
                st["graphs"], st["ptrs"] = {}, ptrs
            else:
                st = self._fused = None
        if st is None:
            dev = fp.flat.device
            st = self._fused = {
                "m": torch.zeros_like(fp.flat), "v": torch.zeros_like(fp.flat),
                "scratch": torch.zeros(_lib.NORM_BLOCKS, device=dev, dtype=torch.float32),
                "mse_scratch": torch.zeros(_lib.NORM_BLOCKS, device=dev, dtype=torch.float32),
                "loss": torch.zeros(1, device=dev, dtype=torch.float32),
                "gnorm": torch.zeros(1, device=dev, dtype=torch.float32),
                "step": torch.zeros(1, device=dev, dtype=torch.int32),
                "lr": torch.full((1,), float(self.opt.lr), device=dev, dtype=torch.float32),
                "lr_host": float(self.opt.lr), "covered": 0, "graphs": {}, "ptrs": ptrs,
                "allreduce": FlatGradAllReduce(),
                "overlap": OverlappedGradAllReduce(wire=getattr(self.opt, "grad_wire", "f32")),
            }
        return st

    def _set_lr(self, lr):
        """Learning rate of the next fused update: a DEVICE scalar the Adam kernel reads, so a captured step follows
        an LR schedule without re-capture.  Written eagerly (never inside a capture), only when it changes."""
        st = self.fused_state()
        lr = float(self.opt.lr if lr is None else lr)
        if lr != st["lr_host"]:
            st["lr"].fill_(lr)
            st["lr_host"] = lr

    def _text_forward(self, clip_out, eot):
        """Text head of the fused step: CLIP features (B, N, W) + EOT indices -> xf_proj, xf_out and the saved
        activations `_text_backward` needs."""
        core = _core(self.encoder)
        xf_proj, xf_out, tsaved = core._launch_text_head(clip_out, eot, training=True)
        return xf_proj, xf_out, (clip_out, eot, xf_out, tsaved)

    def _text_backward(self, tstate, dxf_proj, dxf_out):
        clip_out, eot, xf_out, tsaved = tstate
        _core(self.encoder)._launch_text_head_backward(clip_out, eot, xf_out, tsaved, dxf_out, dxf_proj,
                                                       want_dclip=False, into_flat=True)

    def _fused_fwd_bwd(self, x_start, t, length, xf_proj, xf_out, noise, clip_out=None, eot=None, exchange=None):
        """q_sample -> [text head] -> denoiser forward -> masked MSE -> denoiser backward [-> text head backward]
        into the flat gradient.  With `clip_out` / `eot` the text embeddings are computed here and the text
        head is trained too (the reference's full update); otherwise xf_proj / xf_out are inputs.
        `exchange` (an OverlappedGradAllReduce that has been begun): the gradient all-reduce of each decoder layer is
        started from inside the backward."""
        core = _core(self.encoder)
        L = _lib.lib()
        st = self.fused_state()
        B, T, F = x_start.shape
        tstate = None
        if clip_out is not None:
            xf_proj, xf_out, tstate = self._text_forward(clip_out, eot)
        if noise is None:
            noise = torch.randn_like(x_start)
        x_t = self.diffusion.q_sample(x_start, t, noise=noise)
        pred, saved = core._launch_forward(x_t, t, length, xf_proj, xf_out, training=True)
        dpred = torch.empty_like(pred)
        _lib.check(L.hig_masked_mse(_lib.ptr(pred), _lib.ptr(noise), _lib.ptr(length), B, T, F,
                                    _lib.ptr(st["loss"]), _lib.ptr(dpred), _lib.ptr(st["mse_scratch"]),
                                    _lib.stream_ptr()))
        if exchange is None:
            _, dxp, dxo = core._launch_backward(x_t, t, length, xf_out, saved, dpred, want_dx=False)
        else:
            _, dxp, dxo = core._launch_backward(x_t, t, length, xf_out, saved, dpred, want_dx=False,
                                                layer_hook=exchange.layer_done, comm_stream=exchange.stream)
        if tstate is not None:
            self._text_backward(tstate, dxp, dxo)

    def _fwd_bwd_overlapped_exchange(self, x_start, t, length, xf_proj, xf_out, noise, clip_out, eot):
        """_fused_fwd_bwd with the gradient exchange of layer l issued while layers l-1 ... 0 are still in backward
        (RCCL on a side stream); returns the world size.  Capturable: see OverlappedGradAllReduce."""
        st = self.fused_state()
        core = _core(self.encoder)
        fp = core.flat_params()
        n = self._fused_numel(clip_out is not None)
        ex = st["overlap"]
        want = getattr(self.opt, "grad_wire", "f32")
        if want != ex.wire:      # (the option is read when the fused state is built; changing it later used to be ignored silently)
            raise RuntimeError("opt.grad_wire changed from %r to %r after the fused training state was created: set it before the "
                               "first step (or drop the state: trainer._fused = None)" % (ex.wire, want))
        nsty = 4 if int(getattr(core.dims(1, 1, 1), "two_person", 0)) == 1 else 3   # stylization blocks per layer
        per_layer, tail = fp.layer_buckets(core.num_layers, nsty, core.latent_dim, core.time_embed_dim)
        assert per_layer[-1][1][1] - per_layer[0][1][0] == core.num_layers * nsty * 2 * core.latent_dim * core.time_embed_dim
        ex.begin(fp.grad, per_layer, tail)
        self._fused_fwd_bwd(x_start, t, length, xf_proj, xf_out, noise, clip_out=clip_out, eot=eot, exchange=ex)
        return ex.finish([(fp.core_numel, n)] if n > fp.core_numel else ())

    def _fused_numel(self, with_text):
        """Floats of the flat buffers one fused step covers: the core, or core + text head."""
        core = _core(self.encoder)
        fp = core.flat_params()
        if with_text:
            if not fp.text_params:
                raise RuntimeError("this model's text head is not on the HIP path (text_head='torch', cap_id or an "
                                   "unsupported head dim): the fused step cannot train it")
            clip_net = getattr(core, "clip", None)
            if clip_net is not None and any(p.requires_grad for p in clip_net.parameters()):
                # MotionInteractionTransformer(no_clip=True) trains the CLIP tower through forward()/update()
                # (interaction_transformer.py:424-427); the fused step runs CLIP frozen, outside the step
                raise NotImplementedError("the CLIP text tower is trainable (no_clip=True): the fused step would "
                                          "silently freeze it -- use forward() + update() for this configuration")
        return fp.numel if with_text else fp.core_numel

    def _fused_clip_adam(self, world, with_text=False):
        """g /= world; clip_grad_norm_(0.5); Adam -- one norm pass + one update pass over the flat buffers."""
        core = _core(self.encoder)
        L = _lib.lib()
        st = self.fused_state()
        fp = core.flat_params()
        n = self._fused_numel(with_text)
        _lib.check(L.hig_sumsq_partial(_lib.ptr(fp.grad), n, 1.0 / world, _lib.ptr(st["scratch"]),
                                       _lib.stream_ptr()))
        bf16 = getattr(core, "storage", "f32") == "bf16"
        # bf16 storage: the update kernel also writes bf16(new parameters) into the weight shadow the next forward reads
        # (fp32 master weights, no separate cast pass over the 81 M parameters)
        shadow = fp.shadow16_buffer() if bf16 else None
        _lib.check(L.hig_clip_adam_shadow(_lib.ptr(fp.flat), _lib.ptr(fp.grad), _lib.ptr(st["m"]), _lib.ptr(st["v"]),
                                          n, st["lr_host"], _lib.ptr(st["lr"]), ADAM_BETAS[0], ADAM_BETAS[1], ADAM_EPS,
                                          GRAD_CLIP, 1.0 / world, _lib.ptr(st["scratch"]), _lib.ptr(st["gnorm"]),
                                          _lib.ptr(st["step"]), _lib.ptr(shadow), fp.core_numel if bf16 else 0,
                                          _lib.stream_ptr()))
        st["covered"] = max(st["covered"], n)
        core.params_changed()
        if bf16:
            fp.shadow16_mark_current(core._param_version())

    def train_step_fused(self, x_start, t, length, xf_proj=None, xf_out=None, noise=None, lr=None, clip_out=None,
                         eot=None):
        """One DDPM training step on device tensors, reference semantics (ddpm_trainer.py:97-119,172-187):
          x_t = q_sample(x0, t, noise); pred = denoiser(x_t, t); loss = masked MSE(pred, noise);
          backward; grads = all_reduce(grads) / world; clip_grad_norm_(0.5); Adam.
        With text embeddings (`xf_proj`, `xf_out`) the denoiser-core parameters are updated; with CLIP features
        (`clip_out` (B, N, W) batch-first, `eot` (B,), see MotionTransformer._clip_features) the text head runs
        inside the step and EVERY trainable parameter is updated -- the reference's full update.
        With more than one rank the replicas must start identical: call `sync_replicas()` once (train() does).
        No host synchronisation: the loss stays on the device (`fused_state()['loss']`)."""
        st = self.fused_state()
        with_text = clip_out is not None
        n = self._fused_numel(with_text)
        self._set_lr(lr)
        core = _core(self.encoder)
        fp = core.flat_params()
        if OverlappedGradAllReduce.active() and getattr(self.opt, "overlap_allreduce", True):
            world = self._fwd_bwd_overlapped_exchange(x_start, t, length, xf_proj, xf_out, noise, clip_out, eot)
        else:
            self._fused_fwd_bwd(x_start, t, length, xf_proj, xf_out, noise, clip_out=clip_out, eot=eot)
            world = st["allreduce"](fp.grad[:n])                                  # one all-reduce after the backward
        self._fused_clip_adam(world, with_text)
        return st["loss"]

    def train_fused_batch(self, batch_data, captured=False, noise=None):
        """forward(batch) + update() of the reference (ddpm_trainer.py:97-119,180-187) as ONE fused step over every
        trainable parameter: batch -> device, t ~ sampler, CLIP features -> train_step_fused (eager launches: measured
        faster than the replayed graph at every batch size since the backward runs on two streams), or
        train_step_captured with captured=True (two graph launches per step, no other host work)."""
        caption, x_start, cur_len = self._stage_batch(batch_data)
        t, _ = self.sampler.sample(x_start.shape[0], x_start.device)
        clip_out, eot = self.clip_inputs(caption)
        step = self.train_step_captured if captured else self.train_step_fused
        return step(x_start.contiguous(), t, cur_len, noise=noise, clip_out=clip_out, eot=eot)

    def clip_inputs(self, caption):
        """Captions -> (clip_out (B, N, W) batch-first fp32, eot (B,)) for `train_step_fused(clip_out=...)`: the
        frozen CLIP tower on its own ops, outside the fused / captured region."""
        core = _core(self.encoder)
        tokens, feat = core._clip_features(caption, self.device)
        return feat.permute(1, 0, 2).float().contiguous(), tokens.argmax(dim=-1).contiguous()

    def train_step_captured(self, x_start, t, length, xf_proj=None, xf_out=None, noise=None, lr=None, clip_out=None,
                            eot=None):
        """The same step as hipGraphs.  Captured once per batch shape.  One rank: ONE graph (q_sample [+ text head] +
        forward + loss + backward + clip + Adam).  With a gradient exchange (more than one rank, or HIG_FORCE_EXCHANGE)
        still ONE graph: the per-layer RCCL all-reduces of `OverlappedGradAllReduce` are captured with the backward they
        overlap (RCCL's kernels are graph nodes on the communication stream's branch), so a replay costs the host one
        launch and no per-collective work.  Fallback, taken when `opt.capture_exchange` is False or the capture of the
        collectives fails: graph A = ... + backward, the all-reduce of the flat gradient enqueued eagerly on the same
        stream, graph B = clip + Adam.  Inputs are copied into static device buffers; `t` arrives from the host
        sampler through the same staging copy (the reference draws it with numpy, gaussian_diffusion.py:47-62), the
        learning rate through a device scalar (`_set_lr`).  A graph is keyed on the shapes AND on the flat parameter /
        gradient buffers it was captured against.  Text inputs as in `train_step_fused`: embeddings, or CLIP
        features."""
        st = self.fused_state()
        world = _dist_world()
        split = exchange_active()         # there is an exchange to run (more than one rank, or HIG_FORCE_EXCHANGE)
        # (gloo stages through the host: its collectives cannot be captured -- that backend always takes the split form)
        in_graph = (split and getattr(self.opt, "capture_exchange", True) and getattr(self.opt, "overlap_allreduce", True)
                    and dist.get_backend() == "nccl")
        with_text = clip_out is not None
        text_in = (clip_out, eot) if with_text else (xf_proj, xf_out)
        n = self._fused_numel(with_text)
        self._set_lr(lr)
        key = (tuple(x_start.shape), tuple(text_in[0].shape), tuple(text_in[1].shape), with_text, noise is None,
               world, split, in_graph, st["ptrs"])
        cap = st["graphs"].get(key)
        if cap is None:
            static = {"x0": x_start.clone(), "t": t.clone(), "length": length.clone(),
                      "ta": text_in[0].clone(), "tb": text_in[1].clone(),
                      "noise": None if noise is None else noise.clone()}
            sargs = ((static["x0"], static["t"], static["length"], None, None, static["noise"], static["ta"], static["tb"])
                     if with_text else
                     (static["x0"], static["t"], static["length"], static["ta"], static["tb"], static["noise"], None, None))

            def part_a():
                self._fused_fwd_bwd(*sargs[:6], clip_out=sargs[6], eot=sargs[7])

            def whole_step_with_exchange():
                w = self._fwd_bwd_overlapped_exchange(*sargs)
                self._fused_clip_adam(w, with_text)

            # warm-up (allocations, workspace pools, RCCL's communicator) on a side stream, as torch.cuda.graph requires;
            # it runs a real step, so restore parameters / moments / step counter afterwards
            core = _core(self.encoder)
            fp = core.flat_params()
            keep = (fp.flat.clone(), st["m"].clone(), st["v"].clone(), st["step"].clone(), st["covered"])
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                if in_graph:
                    whole_step_with_exchange()
                else:
                    part_a()
                    self._fused_clip_adam(1, with_text)
            torch.cuda.current_stream().wait_stream(s)

            def restore():
                with torch.no_grad():
                    fp.flat.copy_(keep[0])
                    st["m"].copy_(keep[1])
                    st["v"].copy_(keep[2])
                    st["step"].copy_(keep[3])
                st["covered"] = keep[4]
                core.params_changed()
                if getattr(core, "storage", "f32") == "bf16":   # the warm-up's update also wrote the bf16 shadow: re-derive it
                    fp.shadow16(core._param_version())          # from the restored master weights, eagerly (not inside the capture)
            restore()
            # thread_local: other threads (the RCCL watchdog polls its events) may call into HIP while we capture
            ga, gb = torch.cuda.CUDAGraph(), None
            if in_graph:
                # The warm-up's collectives have finished AND the process group's watchdog thread has retired them before any
                # event is recorded under capture (once, on a loaded box, it queried an event "last recorded in a capturing
                # stream" and took the process down: profiles/r05_notes.md section 5).  Round 6: no Work handle of the exchange
                # survives its call (parallel.OverlappedGradAllReduce._reduce) and the wait is the process group's own
                # waitForPendingWorks instead of a sleep.
                from ..parallel import all_ranks_agree, drain_pending_collectives
                drain_pending_collectives()
                err = None
                try:
                    with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                        whole_step_with_exchange()
                except Exception as e:      # RCCL refused the capture
                    torch.cuda.synchronize()
                    err = "%s: %s" % (type(e).__name__, e)
                # The fallback changes the collectives a rank issues per step (one flat all-reduce instead of 2 L + 2 captured
                # ones): every rank takes it, or none does.
                if not all_ranks_agree(err is None, x_start.device):
                    st.setdefault("capture_exchange_errors", {})[key] = err or "another rank could not capture the exchange"
                    st["capture_exchange_error"] = st["capture_exchange_errors"][key]
                    in_graph = False
                    ga = torch.cuda.CUDAGraph()     # graph A | eager all-reduce | graph B
                    restore()
            if not split:
                with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                    part_a()
                    self._fused_clip_adam(1, with_text)
            elif not in_graph:
                with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                    part_a()
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
                    self._fused_clip_adam(world, with_text)
            cap = st["graphs"][key] = (static, ga, gb)
            st["captured_form"] = "one graph" if not split else ("one graph, exchange inside" if in_graph
                                                                 else "graph A | all-reduce | graph B")
            st.setdefault("captured_forms", {})[key] = st["captured_form"]      # (per graph; "captured_form" = the latest capture)
        static, ga, gb = cap
        with torch.no_grad():
            static["x0"].copy_(x_start)
            static["t"].copy_(t)
            static["length"].copy_(length)
            static["ta"].copy_(text_in[0])
            static["tb"].copy_(text_in[1])
            if noise is not None:
                static["noise"].copy_(noise)
        ga.replay()
        if gb is not None:
            st["allreduce"](_core(self.encoder).flat_params().grad[:n])
            gb.replay()
        st["covered"] = max(st["covered"], n)
        core = _core(self.encoder)
        core.params_changed()
        if getattr(core, "storage", "f32") == "bf16":       # (the replayed update kernel has just written the shadow)
            fp = core.flat_params()
            fp.shadow16_mark_current(core._param_version())
        return st["loss"]
