"""MI355X-native `GaussianDiffusion`: the public surface of the reference class
(codes/models/gaussian_diffusion.py:312-1055, itself OpenAI guided-diffusion) for the branch the
trainers select -- EPSILON / FIXED_SMALL / MSE / linear betas / uniform sampler
(codes/trainers/ddpm_trainer.py:37-46) -- with

  * float64 numpy schedule tables exactly as the reference builds them (:343-380), mirrored once
    per device as an fp32 table (the `.float()` of `_extract_into_tensor`, :1137-1150) instead
    of 8 host->device copies per sampling step;
  * q_sample / the p_sample update as fused HIP kernels (hig_q_sample, hig_p_sample_step);
  * `p_sample_loop` captured as a hipGraph (one denoiser step + update + device-side step
    counter, replayed num_timesteps times) when the model is our MotionTransformer.

The other fixed-variance parametrisations of the reference class (START_X / PREVIOUS_X mean types, FIXED_LARGE
variance, the cosine schedule, q_mean_variance, the eps <-> x0 <-> x_{t-1} conversions) run as plain tensor
arithmetic on whatever device the tensors live on; only the trainers' combination takes the HIP kernels.  Not built
(no reference tool reaches them): learned variances and the KL / VLB losses that train them, DDIM, cond_fn guidance --
the enums keep their member names and those paths raise NotImplementedError instead of silently running something else.
"""
import enum
import math
from abc import ABC, abstractmethod

import numpy as np
import torch as th

from .. import _lib


def create_named_schedule_sampler(name, diffusion):
    """gaussian_diffusion.py:16-27."""
    if name == "uniform":
        return UniformSampler(diffusion)
    raise NotImplementedError(f"unknown schedule sampler: {name}")


class ScheduleSampler(ABC):
    """gaussian_diffusion.py:30-62: host-side importance sampler over timesteps."""

    @abstractmethod
    def weights(self):
        """numpy array of positive weights, one per diffusion step."""

    def sample(self, batch_size, device):
        w = self.weights()
        p = w / np.sum(w)
        indices_np = np.random.choice(len(p), size=(batch_size,), p=p)
        indices = th.from_numpy(indices_np).long().to(device)
        weights_np = 1 / (len(p) * p[indices_np])
        weights = th.from_numpy(weights_np).float().to(device)
        return indices, weights


class UniformSampler(ScheduleSampler):
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    """gaussian_diffusion.py:229-254.  "linear" (what the reference trainers ask for): betas from 1e-4 to 2e-2 at 1000
    steps, end points scaled by 1000 / N otherwise; "cosine": the alpha-bar curve cos^2((t + 0.008) / 1.008 * pi / 2)
    discretised by betas_for_alpha_bar.  float64."""
    n = num_diffusion_timesteps
    if schedule_name == "linear":
        scale = 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(n, lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    """gaussian_diffusion.py:256-274: beta_i = 1 - alpha_bar((i + 1) / N) / alpha_bar(i / N), capped at max_beta
    (alpha_bar: cumulative product of (1 - beta) as a function of t in [0, 1])."""
    n = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)])


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self == LossType.KL or self == LossType.RESCALED_KL


def mean_flat(tensor):
    return tensor.mean(dim=list(range(1, len(tensor.shape))))


_TAB_ORDER = ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
              "posterior_log_variance_clipped")


def _unwrap(model):
    return getattr(model, "module", model)


class GaussianDiffusion:
    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type, rescale_timesteps=False):
        if model_var_type in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE) or loss_type.is_vb():
            # learned variances need a model with 2 C output channels and the VLB terms that train them: no reference
            # tool builds one (ddpm_trainer.py:40-45 is EPSILON / FIXED_SMALL / MSE)
            raise NotImplementedError("GaussianDiffusion: learned variances and the KL / VLB losses are not built, got "
                                      "%s / %s / %s" % (model_mean_type, model_var_type, loss_type))
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        self.rescale_timesteps = rescale_timesteps

        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        assert len(betas.shape) == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.num_timesteps = int(betas.shape[0])

        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        assert self.alphas_cumprod_prev.shape == (self.num_timesteps,)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(
            np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)

        self.use_hip_graph = True      # p_sample_loop: capture the step and replay it
        self._dev_tabs = {}
        self._debug_zero_noise = False  # tests: deterministic graph-vs-eager comparison

    # ---- device-resident tables -------------------------------------------------------------
    def device_table(self, device):
        """(7, num_timesteps) fp32 table in hig.h order, built once per device."""
        key = str(device)
        if key not in self._dev_tabs:
            tab = np.stack([getattr(self, n) for n in _TAB_ORDER]).astype(np.float32)
            self._dev_tabs[key] = th.from_numpy(tab).to(device).contiguous()
        return self._dev_tabs[key]

    def _fused_ok(self, *tensors):
        return all(t.is_cuda and t.dtype == th.float32 for t in tensors)

    # ---- q(x_t | x_0) ---------------------------------------------------------------------
    def q_mean_variance(self, x_start, t):
        """(mean, variance, log variance) of q(x_t | x_0), each of x_start's shape (gaussian_diffusion.py:382-397)."""
        shape = x_start.shape
        return (_extract_into_tensor(self.sqrt_alphas_cumprod, t, shape) * x_start,
                _extract_into_tensor(1.0 - self.alphas_cumprod, t, shape),
                _extract_into_tensor(self.log_one_minus_alphas_cumprod, t, shape))

    def q_sample(self, x_start, t, noise=None):
        """gaussian_diffusion.py:399-417."""
        if noise is None:
            noise = th.randn_like(x_start)
        assert noise.shape == x_start.shape
        if self._fused_ok(x_start, noise) and not x_start.requires_grad and not noise.requires_grad:
            x0, nz = x_start.contiguous(), noise.contiguous()
            out = th.empty_like(x0)
            B = x0.shape[0]
            _lib.check(_lib.lib().hig_q_sample(_lib.ptr(x0), _lib.ptr(nz), _lib.ptr(t.long().contiguous()),
                                              _lib.ptr(self.device_table(x0.device)), self.num_timesteps, B,
                                              x0.numel() // B, _lib.ptr(out), _lib.stream_ptr()))
            return out
        return (_extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + _extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def q_posterior_mean_variance(self, x_start, x_t, t):
        """gaussian_diffusion.py:419-441."""
        assert x_start.shape == x_t.shape
        posterior_mean = (_extract_into_tensor(self.posterior_mean_coef1, t, x_t.shape) * x_start
                          + _extract_into_tensor(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        posterior_variance = _extract_into_tensor(self.posterior_variance, t, x_t.shape)
        posterior_log_variance_clipped = _extract_into_tensor(self.posterior_log_variance_clipped, t, x_t.shape)
        return posterior_mean, posterior_variance, posterior_log_variance_clipped

    # ---- p(x_{t-1} | x_t) -------------------------------------------------------------------
    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """Model output -> {'mean', 'variance', 'log_variance', 'pred_xstart'} of p(x_{t-1} | x_t) for the fixed
        variances (gaussian_diffusion.py:443-537).  The x0 estimate goes through denoised_fn, then the clamp; for
        START_X / EPSILON the mean is the posterior q(x_{t-1} | x_t, x0_hat), for PREVIOUS_X it is the output itself."""
        B = x.shape[0]
        assert t.shape == (B,)
        out = model(x, self._scale_timesteps(t), **(model_kwargs or {}))
        if self.model_var_type == ModelVarType.FIXED_LARGE:
            # beta_t, with the first entry taken from the posterior variance (better decoder likelihood at t = 0)
            var_tab = np.append(self.posterior_variance[1], self.betas[1:])
            variance = _extract_into_tensor(var_tab, t, x.shape)
            log_variance = _extract_into_tensor(np.log(var_tab), t, x.shape)
        else:
            variance = _extract_into_tensor(self.posterior_variance, t, x.shape)
            log_variance = _extract_into_tensor(self.posterior_log_variance_clipped, t, x.shape)

        def finish(x0):
            if denoised_fn is not None:
                x0 = denoised_fn(x0)
            return x0.clamp(-1, 1) if clip_denoised else x0

        if self.model_mean_type == ModelMeanType.PREVIOUS_X:
            pred_xstart = finish(self._predict_xstart_from_xprev(x_t=x, t=t, xprev=out))
            mean = out
        else:
            pred_xstart = finish(out if self.model_mean_type == ModelMeanType.START_X
                                 else self._predict_xstart_from_eps(x_t=x, t=t, eps=out))
            mean = self.q_posterior_mean_variance(x_start=pred_xstart, x_t=x, t=t)[0]
        assert mean.shape == log_variance.shape == pred_xstart.shape == x.shape
        return {"mean": mean, "variance": variance, "log_variance": log_variance, "pred_xstart": pred_xstart}

    def _predict_xstart_from_eps(self, x_t, t, eps):
        assert x_t.shape == eps.shape
        return (_extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - _extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * eps)

    def _predict_xstart_from_xprev(self, x_t, t, xprev):
        """Posterior mean solved for x0: (x_{t-1} - coef2 x_t) / coef1 (gaussian_diffusion.py:546-554)."""
        assert x_t.shape == xprev.shape
        return (_extract_into_tensor(1.0 / self.posterior_mean_coef1, t, x_t.shape) * xprev
                - _extract_into_tensor(self.posterior_mean_coef2 / self.posterior_mean_coef1, t, x_t.shape) * x_t)

    def _predict_eps_from_xstart(self, x_t, t, pred_xstart):
        """Inverse of _predict_xstart_from_eps (gaussian_diffusion.py:556-560)."""
        return ((_extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - pred_xstart)
                / _extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape))

    def _scale_timesteps(self, t):
        if self.rescale_timesteps:
            return t.float() * (1000.0 / self.num_timesteps)
        return t

    def _is_trainer_branch(self, clip_denoised, denoised_fn, cond_fn, pre_seq, transl_req):
        return (self.model_mean_type == ModelMeanType.EPSILON
                and self.model_var_type == ModelVarType.FIXED_SMALL and not clip_denoised
                and denoised_fn is None and cond_fn is None and pre_seq is None and transl_req is None
                and not self.rescale_timesteps)

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, pre_seq=None,
                 transl_req=None, model_kwargs=None):
        """gaussian_diffusion.py:606-666.  On the trainers' branch the whole update
        (x0-hat, posterior mean, noise) is ONE kernel; other branches use tensor ops."""
        if pre_seq is not None or transl_req is not None:
            raise NotImplementedError("pre_seq / transl_req conditioning is never used by the reference tools")
        if cond_fn is not None:
            raise NotImplementedError("cond_fn guidance is never used by the reference tools")
        if self._is_trainer_branch(clip_denoised, denoised_fn, cond_fn, pre_seq, transl_req) and self._fused_ok(x):
            eps = model(x, t, **(model_kwargs or {}))
            noise = th.randn_like(x)
            xc, ec = x.contiguous(), eps.float().contiguous()
            sample, pred = th.empty_like(xc), th.empty_like(xc)
            B = xc.shape[0]
            _lib.check(_lib.lib().hig_p_sample_step(
                _lib.ptr(xc), _lib.ptr(ec), _lib.ptr(noise.contiguous()), _lib.ptr(t.long().contiguous()),
                _lib.ptr(self.device_table(xc.device)), self.num_timesteps, B, xc.numel() // B,
                _lib.ptr(sample), _lib.ptr(pred), _lib.stream_ptr()))
            return {"sample": sample, "pred_xstart": pred}
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                   model_kwargs=model_kwargs)
        noise = th.randn_like(x)
        nonzero_mask = (t != 0).float().view(-1, *([1] * (len(x.shape) - 1)))
        sample = out["mean"] + nonzero_mask * th.exp(0.5 * out["log_variance"]) * noise
        return {"sample": sample, "pred_xstart": out["pred_xstart"]}

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, pre_seq=None, transl_req=None, progress=False):
        """gaussian_diffusion.py:668-716."""
        core = _unwrap(model)
        if (self.use_hip_graph and hasattr(core, "_launch_forward") and model_kwargs is not None
                and model_kwargs.get("xf_proj") is not None and model_kwargs.get("xf_out") is not None
                and self._is_trainer_branch(clip_denoised, denoised_fn, cond_fn, pre_seq, transl_req)):
            return self._p_sample_loop_graph(core, shape, noise, model_kwargs, device)
        final = None
        for sample in self.p_sample_loop_progressive(
                model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                cond_fn=cond_fn, model_kwargs=model_kwargs, device=device, pre_seq=pre_seq,
                transl_req=transl_req, progress=progress):
            final = sample
        return final["sample"]

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                  cond_fn=None, model_kwargs=None, device=None, pre_seq=None,
                                  transl_req=None, progress=False):
        """gaussian_diffusion.py:718-769: t = N-1 ... 0, fresh noise every step, no_grad."""
        if device is None:
            device = next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else th.randn(*shape, device=device)
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        for i in indices:
            t = th.tensor([i] * shape[0], device=device)
            with th.no_grad():
                out = self.p_sample(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                    cond_fn=cond_fn, model_kwargs=model_kwargs, pre_seq=pre_seq,
                                    transl_req=transl_req)
                yield out
                img = out["sample"]

    def _p_sample_loop_graph(self, core, shape, noise, model_kwargs, device):
        """The 1000-step loop as ONE captured hipGraph replayed num_timesteps times.

        Captured per step: denoiser forward (text context hoisted: it is step-invariant), fresh
        Gaussian noise (torch's graph-safe Philox), the fused x_{t-1} update in place, and the
        device-side `t -= 1`.  Nothing crosses PCIe inside the loop (the reference does ~8 H2D
        copies and B device syncs per step)."""
        if device is None:
            device = next(core.parameters()).device
        B = shape[0]
        with th.no_grad():
            img = (noise.to(device).float().clone() if noise is not None
                   else th.randn(*shape, device=device)).contiguous()
            xf_proj = model_kwargs["xf_proj"].detach().float().contiguous()
            xf_out = model_kwargs["xf_out"].detach().float().contiguous()
            length = model_kwargs.get("length")
            if length is None:
                length = th.full((B,), shape[1], dtype=th.int64, device=device)
            else:
                length = th.as_tensor(length).to(device).long().contiguous()
            t_dev = th.full((B,), self.num_timesteps - 1, dtype=th.int64, device=device)
            z = th.zeros_like(img)
            tab = self.device_table(device)
            L = _lib.lib()
            per = img.numel() // B

            def step():
                eps, _ = core._launch_forward(img, t_dev, length, xf_proj, xf_out, training=False)
                if not self._debug_zero_noise:
                    z.normal_()
                _lib.check(L.hig_p_sample_step(_lib.ptr(img), _lib.ptr(eps), _lib.ptr(z), _lib.ptr(t_dev),
                                               _lib.ptr(tab), self.num_timesteps, B, per, _lib.ptr(img),
                                               None, _lib.stream_ptr()))
                _lib.check(L.hig_dec_timesteps(_lib.ptr(t_dev), B, _lib.stream_ptr()))

            # warm-up on a side stream (allocations, text context), then undo its effect
            img0 = img.clone()
            s = th.cuda.Stream()
            s.wait_stream(th.cuda.current_stream())
            with th.cuda.stream(s):
                step()
            th.cuda.current_stream().wait_stream(s)
            img.copy_(img0)
            t_dev.fill_(self.num_timesteps - 1)
            graph = th.cuda.CUDAGraph()
            with th.cuda.graph(graph, capture_error_mode="thread_local"):
                step()
            # capture does not execute: state is still (img0, N-1)
            for _ in range(self.num_timesteps):
                graph.replay()
        return img

    # ---- training ---------------------------------------------------------------------------
    def training_losses(self, model, x_start, t, model_kwargs=None, noise=None, forward_twice=False):
        """gaussian_diffusion.py:978-1055, MSE branch: {'mse', 'target', 'pred'}."""
        if model_kwargs is None:
            model_kwargs = {}
        if noise is None:
            noise = th.randn_like(x_start)
        x_t = self.q_sample(x_start, t, noise=noise)
        if forward_twice:
            B = x_t.size(0) // 2
            x_t = th.cat([x_t[:B], x_t[:B], x_t[B:], x_t[B:]])
            t = th.cat([t, t])
            x_start = th.cat([x_start[:B], x_start[:B], x_start[B:], x_start[B:]])
            noise = th.cat([noise[:B], noise[:B], noise[B:], noise[B:]])
        model_output = model(x_t, self._scale_timesteps(t), **model_kwargs)
        # regression target by what the model predicts: the injected noise (the trainers' choice), x_0, or the
        # posterior mean of x_{t-1}                                                 (gaussian_diffusion.py:1041-1047)
        if self.model_mean_type == ModelMeanType.EPSILON:
            target = noise
        elif self.model_mean_type == ModelMeanType.START_X:
            target = x_start
        else:
            target = self.q_posterior_mean_variance(x_start=x_start, x_t=x_t, t=t)[0]
        assert model_output.shape == target.shape == x_start.shape
        return {"mse": mean_flat((target - model_output) ** 2).view(-1, 1).mean(-1), "target": target,
                "pred": model_output}

    # ---- names kept for API compatibility; never reached by the reference tools --------------
    def ddim_sample(self, *a, **k):
        raise NotImplementedError("DDIM sampling is not used by any reference tool")

    ddim_sample_loop = ddim_reverse_sample = ddim_sample_loop_progressive = ddim_sample

    def calc_bpd_loop(self, *a, **k):
        raise NotImplementedError("VLB / bpd evaluation is not used by any reference tool")


def _extract_into_tensor(arr, timesteps, broadcast_shape):
    """gaussian_diffusion.py:1137-1150."""
    res = th.from_numpy(np.asarray(arr)).to(device=timesteps.device)[timesteps].float()
    while len(res.shape) < len(broadcast_shape):
        res = res[..., None]
    return res.expand(broadcast_shape)
