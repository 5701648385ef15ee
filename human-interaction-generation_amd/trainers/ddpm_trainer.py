"""MI355X-native `DDPMTrainer`: the reference trainer's public surface
(codes/trainers/ddpm_trainer.py:29-266) over the HIP denoiser.

Two ways to take a training step:
  * `forward(batch)` + `update()` -- the reference's own sequence (autograd, torch Adam,
    clip_grad_norm_), kept so existing driver code runs unchanged;
  * `train_step_fused(...)` -- the MI355X path: explicit forward/backward launches, masked-MSE,
    ONE flat-buffer RCCL all-reduce and a fused clip+Adam kernel, hipGraph-capturable.
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
from ..parallel import FlatGradAllReduce, ShardedSampler


def _core(encoder):
    return getattr(encoder, "module", encoder)


class DDPMTrainer(object):

    def __init__(self, args, encoder):
        self.opt = args
        self.device = args.device
        self.encoder = encoder
        self.multi = False
        self.diffusion_steps = args.diffusion_steps
        sampler = 'uniform'
        beta_scheduler = 'linear'
        betas = get_named_beta_schedule(beta_scheduler, self.diffusion_steps)
        self.diffusion = GaussianDiffusion(
            betas=betas,
            model_mean_type=ModelMeanType.EPSILON,
            model_var_type=ModelVarType.FIXED_SMALL,
            loss_type=LossType.MSE
        )
        self.sampler = create_named_schedule_sampler(sampler, self.diffusion)
        self.sampler_name = sampler
        if args.is_train:
            self.mse_criterion = torch.nn.MSELoss(reduction='none')
        self.to(self.device)
        self._fused = None

    @staticmethod
    def zero_grad(opt_list):
        for opt in opt_list:
            opt.zero_grad()

    @staticmethod
    def clip_norm(network_list):
        for network in network_list:
            clip_grad_norm_(network.parameters(), 0.5)

    @staticmethod
    def step(opt_list):
        for opt in opt_list:
            opt.step()

    def forward(self, batch_data, eval_mode=False):
        """ddpm_trainer.py:97-119."""
        caption, motions, m_lens = batch_data
        motions = motions.detach().to(self.device).float()
        self.caption = caption
        self.motions = motions
        x_start = motions
        B, T = x_start.shape[:2]
        cur_len = torch.LongTensor([min(T, int(m_len)) for m_len in m_lens]).to(self.device)
        t, _ = self.sampler.sample(B, x_start.device)
        output = self.diffusion.training_losses(
            model=self.encoder,
            x_start=x_start,
            t=t,
            model_kwargs={"text": caption, "length": cur_len}
        )
        self.real_noise = output['target']
        self.fake_noise = output['pred']
        self.src_mask = _core(self.encoder).generate_src_mask(T, cur_len).to(x_start.device)

    def generate_batch(self, caption, m_lens, dim_pose):
        """ddpm_trainer.py:121-150: text encoded once, then the (graph-captured) sampling loop."""
        xf_proj, xf_out = _core(self.encoder).encode_text(caption, self.device)
        B = len(caption)
        T = min(int(m_lens.max()), _core(self.encoder).num_frames)
        output = self.diffusion.p_sample_loop(
            self.encoder,
            (B, T, dim_pose),
            clip_denoised=False,
            progress=True,
            model_kwargs={
                'xf_proj': xf_proj,
                'xf_out': xf_out,
                'length': m_lens
            })
        return output

    def generate(self, caption, m_lens, dim_pose, batch_size=1024):
        """ddpm_trainer.py:152-170."""
        N = len(caption)
        cur_idx = 0
        self.encoder.eval()
        all_output = []
        while cur_idx < N:
            if cur_idx + batch_size >= N:
                batch_caption = caption[cur_idx:]
                batch_m_lens = m_lens[cur_idx:]
            else:
                batch_caption = caption[cur_idx: cur_idx + batch_size]
                batch_m_lens = m_lens[cur_idx: cur_idx + batch_size]
            output = self.generate_batch(batch_caption, batch_m_lens, dim_pose)
            B = output.shape[0]
            for i in range(B):
                all_output.append(output[i])
            cur_idx += batch_size
        return all_output

    def backward_G(self):
        """ddpm_trainer.py:172-178."""
        loss_mot_rec = self.mse_criterion(self.fake_noise, self.real_noise).mean(dim=-1)
        loss_mot_rec = (loss_mot_rec * self.src_mask).sum() / self.src_mask.sum()
        self.loss_mot_rec = loss_mot_rec
        loss_logs = OrderedDict({})
        loss_logs['loss_mot_rec'] = self.loss_mot_rec.item()
        return loss_logs

    def update(self):
        """ddpm_trainer.py:180-187."""
        self.zero_grad([self.opt_encoder])
        loss_logs = self.backward_G()
        self.loss_mot_rec.backward()
        self.clip_norm([self.encoder])
        self.step([self.opt_encoder])
        return loss_logs

    def to(self, device):
        if self.opt.is_train:
            self.mse_criterion.to(device)
        self.encoder = self.encoder.to(device)

    def train_mode(self):
        self.encoder.train()

    def eval_mode(self):
        self.encoder.eval()

    def save(self, file_name, ep, total_it):
        """ddpm_trainer.py:200-211: keys opt_encoder / ep / total_it / encoder."""
        state = {
            'opt_encoder': self.opt_encoder.state_dict(),
            'ep': ep,
            'total_it': total_it,
            'encoder': _core(self.encoder).state_dict(),
        }
        torch.save(state, file_name)

    def load(self, model_dir):
        checkpoint = torch.load(model_dir, map_location=self.device)
        if self.opt.is_train:
            self.opt_encoder.load_state_dict(checkpoint['opt_encoder'])
        _core(self.encoder).load_state_dict(checkpoint['encoder'], strict=True)
        return checkpoint['ep'], checkpoint.get('total_it', 0)

    def train(self, train_dataset, rank, world_size):
        """ddpm_trainer.py:220-266 (epoch loop, logging, checkpoint cadence)."""
        self.to(self.device)
        self.opt_encoder = optim.Adam(self.encoder.parameters(), lr=self.opt.lr)
        it = 0
        cur_epoch = 0
        if self.opt.is_continue:
            model_dir = pjoin(self.opt.model_dir, 'latest.tar')
            cur_epoch, it = self.load(model_dir)
        start_time = time.time()
        sampler = ShardedSampler(len(train_dataset), rank, world_size, shuffle=True)
        train_loader = torch.utils.data.DataLoader(
            train_dataset, batch_size=self.opt.batch_size, sampler=sampler, drop_last=True,
            num_workers=getattr(self.opt, "num_workers", 4), shuffle=False)
        logs = OrderedDict()
        for epoch in range(cur_epoch, self.opt.num_epochs):
            self.train_mode()
            for i, batch_data in enumerate(train_loader):
                if getattr(self.opt, "fused_step", False):
                    # MI355X path: the same update as forward() + update() as one fused step (`opt.fused_graph`:
                    # replayed as a hipGraph -- no host work per step, but the backward's second stream overlaps less)
                    log_dict = OrderedDict({'loss_mot_rec': self.train_fused_batch(
                        batch_data, captured=getattr(self.opt, "fused_graph", False)).item()})
                else:
                    self.forward(batch_data)
                    log_dict = self.update()
                for k, v in log_dict.items():
                    logs[k] = logs.get(k, 0) + v
                it += 1
                if it % self.opt.log_every == 0 and rank == 0:
                    mean_loss = OrderedDict({tag: value / self.opt.log_every for tag, value in logs.items()})
                    logs = OrderedDict()
                    msg = ' '.join('%s: %.4f' % kv for kv in mean_loss.items())
                    print('epoch: %3d niter: %6d inner_iter: %4d %.0fs %s'
                          % (epoch, it, i, time.time() - start_time, msg))
                if it % self.opt.save_latest == 0 and rank == 0:
                    self.save(pjoin(self.opt.model_dir, 'latest.tar'), epoch, it)
            if rank == 0:
                self.save(pjoin(self.opt.model_dir, 'latest.tar'), epoch, it)
            if epoch % self.opt.save_every_e == 0 and rank == 0:
                self.save(pjoin(self.opt.model_dir, 'ckpt_e%03d.tar' % (epoch)), epoch, total_it=it)

    # ------------------------------------------------------------------------------------------
    # MI355X fused step
    # ------------------------------------------------------------------------------------------
    def fused_state(self):
        """Flat Adam moments, loss / norm scalars and scratch for the fused step (lazy)."""
        if self._fused is None:
            core = _core(self.encoder)
            fp = core.flat_params()
            fp.ensure_grad()
            dev = fp.flat.device
            self._fused = {
                "m": torch.zeros_like(fp.flat), "v": torch.zeros_like(fp.flat),
                "scratch": torch.zeros(_lib.NORM_BLOCKS, device=dev, dtype=torch.float32),
                "mse_scratch": torch.zeros(_lib.NORM_BLOCKS, device=dev, dtype=torch.float32),
                "loss": torch.zeros(1, device=dev, dtype=torch.float32),
                "gnorm": torch.zeros(1, device=dev, dtype=torch.float32),
                "step": torch.zeros(1, device=dev, dtype=torch.int32),
                "allreduce": FlatGradAllReduce(),
            }
        return self._fused

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

    def _fused_fwd_bwd(self, x_start, t, length, xf_proj, xf_out, noise, clip_out=None, eot=None):
        """q_sample -> [text head] -> denoiser forward -> masked MSE -> denoiser backward [-> text head backward]
        into the flat gradient.  With `clip_out` / `eot` the text embeddings are computed here and the text
        head is trained too (the reference's full update); otherwise xf_proj / xf_out are inputs."""
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
        _, dxp, dxo = core._launch_backward(x_t, t, length, xf_out, saved, dpred, want_dx=False)
        if tstate is not None:
            self._text_backward(tstate, dxp, dxo)

    def _fused_numel(self, with_text):
        """Floats of the flat buffers one fused step covers: the core, or core + text head."""
        fp = _core(self.encoder).flat_params()
        if with_text and not fp.text_params:
            raise RuntimeError("this model's text head is not on the HIP path (text_head='torch', cap_id or an "
                               "unsupported head dim): the fused step cannot train it")
        return fp.numel if with_text else fp.core_numel

    def _fused_clip_adam(self, world, lr, with_text=False):
        """g /= world; clip_grad_norm_(0.5); Adam -- one norm pass + one update pass over the flat buffers."""
        core = _core(self.encoder)
        L = _lib.lib()
        st = self.fused_state()
        fp = core.flat_params()
        n = self._fused_numel(with_text)
        _lib.check(L.hig_sumsq_partial(_lib.ptr(fp.grad), n, 1.0 / world, _lib.ptr(st["scratch"]),
                                       _lib.stream_ptr()))
        _lib.check(L.hig_clip_adam(_lib.ptr(fp.flat), _lib.ptr(fp.grad), _lib.ptr(st["m"]), _lib.ptr(st["v"]),
                                   n, float(lr if lr is not None else self.opt.lr), 0.9, 0.999, 1e-8, 0.5,
                                   1.0 / world, _lib.ptr(st["scratch"]), _lib.ptr(st["gnorm"]),
                                   _lib.ptr(st["step"]), _lib.stream_ptr()))
        core._textctx_cache = None  # parameters changed under the cached text context

    def train_step_fused(self, x_start, t, length, xf_proj=None, xf_out=None, noise=None, lr=None, clip_out=None,
                         eot=None):
        """One DDPM training step on device tensors, reference semantics (ddpm_trainer.py:97-119,172-187):
          x_t = q_sample(x0, t, noise); pred = denoiser(x_t, t); loss = masked MSE(pred, noise);
          backward; grads = all_reduce(grads) / world; clip_grad_norm_(0.5); Adam.
        With text embeddings (`xf_proj`, `xf_out`) the denoiser-core parameters are updated; with CLIP features
        (`clip_out` (B, N, W) batch-first, `eot` (B,), see MotionTransformer._clip_features) the text head runs
        inside the step and EVERY trainable parameter is updated -- the reference's full update.
        No host synchronisation: the loss stays on the device (`fused_state()['loss']`)."""
        st = self.fused_state()
        with_text = clip_out is not None
        n = self._fused_numel(with_text)
        self._fused_fwd_bwd(x_start, t, length, xf_proj, xf_out, noise, clip_out=clip_out, eot=eot)
        world = st["allreduce"](_core(self.encoder).flat_params().grad[:n])      # sum over ranks (RCCL, xGMI)
        self._fused_clip_adam(world, lr, with_text)
        return st["loss"]

    def train_fused_batch(self, batch_data, captured=False, noise=None):
        """forward(batch) + update() of the reference (ddpm_trainer.py:97-119,180-187) as ONE fused step over every
        trainable parameter: batch -> device, t ~ sampler, CLIP features -> train_step_fused (eager launches: measured
        faster than the replayed graph at every batch size since the backward runs on two streams), or
        train_step_captured with captured=True (two graph launches per step, no other host work)."""
        caption, motions, m_lens = batch_data
        x_start = motions.detach().to(self.device).float().contiguous()
        B, T = x_start.shape[:2]
        cur_len = torch.LongTensor([min(T, int(m_len)) for m_len in m_lens]).to(self.device)
        t, _ = self.sampler.sample(B, x_start.device)
        clip_out, eot = self.clip_inputs(caption)
        step = self.train_step_captured if captured else self.train_step_fused
        return step(x_start, t, cur_len, noise=noise, clip_out=clip_out, eot=eot)

    def clip_inputs(self, caption):
        """Captions -> (clip_out (B, N, W) batch-first fp32, eot (B,)) for `train_step_fused(clip_out=...)`: the
        frozen CLIP tower on its own ops, outside the fused / captured region."""
        core = _core(self.encoder)
        tokens, feat = core._clip_features(caption, self.device)
        return feat.permute(1, 0, 2).float().contiguous(), tokens.argmax(dim=-1).contiguous()

    def train_step_captured(self, x_start, t, length, xf_proj=None, xf_out=None, noise=None, lr=None, clip_out=None,
                            eot=None):
        """The same step as hipGraphs.  Captured once per batch shape: graph A = q_sample [+ text head] + forward
        + loss + backward, graph B = clip + Adam; the RCCL all-reduce of the flat gradient runs between them on
        the same stream (world == 1: A and B are one graph).  Inputs are copied into static device buffers, so
        the host cost per step is two graph launches whatever the ~600 kernels inside; `t` arrives from the
        host sampler through the same staging copy (the reference draws it with numpy,
        gaussian_diffusion.py:47-62).  Text inputs as in `train_step_fused`: embeddings, or CLIP features."""
        st = self.fused_state()
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        with_text = clip_out is not None
        text_in = (clip_out, eot) if with_text else (xf_proj, xf_out)
        n = self._fused_numel(with_text)
        key = (tuple(x_start.shape), tuple(text_in[0].shape), tuple(text_in[1].shape), with_text, noise is None,
               world, lr)
        cap = st.setdefault("graphs", {}).get(key)
        if cap is None:
            static = {"x0": x_start.clone(), "t": t.clone(), "length": length.clone(),
                      "ta": text_in[0].clone(), "tb": text_in[1].clone(),
                      "noise": None if noise is None else noise.clone()}

            def part_a():
                if with_text:
                    self._fused_fwd_bwd(static["x0"], static["t"], static["length"], None, None, static["noise"],
                                        clip_out=static["ta"], eot=static["tb"])
                else:
                    self._fused_fwd_bwd(static["x0"], static["t"], static["length"], static["ta"], static["tb"],
                                        static["noise"])

            # warm-up (allocations, workspace pools) on a side stream, as torch.cuda.graph requires;
            # it runs a real step, so restore parameters / moments / step counter afterwards
            core = _core(self.encoder)
            fp = core.flat_params()
            keep = (fp.flat.clone(), st["m"].clone(), st["v"].clone(), st["step"].clone())
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                part_a()
                self._fused_clip_adam(1, lr, with_text)
            torch.cuda.current_stream().wait_stream(s)
            with torch.no_grad():
                fp.flat.copy_(keep[0])
                st["m"].copy_(keep[1])
                st["v"].copy_(keep[2])
                st["step"].copy_(keep[3])
            # thread_local: other threads (the RCCL watchdog polls its events) may call into HIP while we capture
            ga, gb = torch.cuda.CUDAGraph(), None
            if world == 1:
                with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                    part_a()
                    self._fused_clip_adam(1, lr, with_text)
            else:
                with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                    part_a()
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
                    self._fused_clip_adam(world, lr, with_text)
            cap = st["graphs"][key] = (static, ga, gb)
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
        _core(self.encoder)._textctx_cache = None
        return st["loss"]
