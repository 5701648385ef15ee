"""MI355X-native `MotionTransformer`: same constructor, attributes, state-dict keys and call
signature as the reference module (codes/models/transformer.py:288-426), with the per-timestep
denoiser math running in hand-written HIP kernels behind the C ABI of include/hig.h.

`encode_text`: CLIP itself (frozen; stubbed when the `clip` package is absent) runs on
its own PyTorch ops; the trainable head behind it -- text_pre_proj, the 4-layer post-norm encoder, text_ln,
the EOT gather and text_proj -- goes through hig_text_head_fwd / _bwd (`text_head="torch"` keeps it on stock
ops, e.g. for head dims the kernels do not cover).  The nn.Modules of the head are parameter containers.

There is no CPU execution path: calling the module on host tensors, or without libhig.so, raises.
"""
import ctypes as C
import math

import torch
from torch import nn

from .. import _lib

try:  # the real OpenAI CLIP when installed (transformer.py:10), else the deterministic stand-in
    import clip  # type: ignore
except Exception:  # pragma: no cover - depends on the environment
    from . import stub_clip as clip


def timestep_embedding(timesteps, dim, max_period=10000):
    """Sinusoidal embedding, cos first (transformer.py:15-32).  Device tensors go through the
    HIP kernel; kept as a module-level function because reference callers import it."""
    if timesteps.is_cuda:
        t = timesteps.long().contiguous()
        out = torch.empty(t.shape[0], dim, device=t.device, dtype=torch.float32)
        _lib.check(_lib.lib().hig_timestep_embedding(_lib.ptr(t), t.shape[0], dim, _lib.ptr(out),
                                                    _lib.stream_ptr()))
        return out
    raise RuntimeError("timestep_embedding: ROCm device tensor required (no CPU fallback)")


def set_requires_grad(nets, requires_grad=False):
    """transformer.py:35-48."""
    if not isinstance(nets, list):
        nets = [nets]
    for net in nets:
        if net is not None:
            for param in net.parameters():
                param.requires_grad = requires_grad


def zero_module(module):
    """transformer.py:51-57."""
    for p in module.parameters():
        p.detach().zero_()
    return module


# --- parameter containers: same submodule / key names as the reference; math lives in HIP -----
class StylizationBlock(nn.Module):
    """Parameters of transformer.py:60-73 (emb_layers.1, norm, out_layers.2 [zero-init])."""

    def __init__(self, latent_dim, time_embed_dim, dropout):
        super().__init__()
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(time_embed_dim, 2 * latent_dim))
        self.norm = nn.LayerNorm(latent_dim)
        self.out_layers = nn.Sequential(nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(nn.Linear(latent_dim, latent_dim)))


class _SelfAttention(nn.Module):
    """Parameters of (Linear)TemporalSelfAttention, transformer.py:91-99 / 198-206."""

    def __init__(self, latent_dim, num_head, dropout, time_embed_dim):
        super().__init__()
        self.num_head = num_head
        self.norm = nn.LayerNorm(latent_dim)
        self.query = nn.Linear(latent_dim, latent_dim)
        self.key = nn.Linear(latent_dim, latent_dim)
        self.value = nn.Linear(latent_dim, latent_dim)
        self.dropout = nn.Dropout(dropout)
        self.proj_out = StylizationBlock(latent_dim, time_embed_dim, dropout)


class _CrossAttention(nn.Module):
    """Parameters of (Linear)TemporalCrossAttention, transformer.py:124-133 / 231-240."""

    def __init__(self, latent_dim, text_latent_dim, num_head, dropout, time_embed_dim):
        super().__init__()
        self.num_head = num_head
        self.norm = nn.LayerNorm(latent_dim)
        self.text_norm = nn.LayerNorm(text_latent_dim)
        self.query = nn.Linear(latent_dim, latent_dim)
        self.key = nn.Linear(text_latent_dim, latent_dim)
        self.value = nn.Linear(text_latent_dim, latent_dim)
        self.dropout = nn.Dropout(dropout)
        self.proj_out = StylizationBlock(latent_dim, time_embed_dim, dropout)


class FFN(nn.Module):
    """Parameters of transformer.py:157-165 (linear2 zero-init)."""

    def __init__(self, latent_dim, ffn_dim, dropout, time_embed_dim):
        super().__init__()
        self.linear1 = nn.Linear(latent_dim, ffn_dim)
        self.linear2 = zero_module(nn.Linear(ffn_dim, latent_dim))
        self.activation = nn.GELU()
        self.dropout = nn.Dropout(dropout)
        self.proj_out = StylizationBlock(latent_dim, time_embed_dim, dropout)


class _DecoderLayer(nn.Module):
    """transformer.py:173-194 / 264-285: sa_block -> ca_block -> ffn."""

    def __init__(self, latent_dim, text_latent_dim, time_embed_dim, ffn_dim, num_head, dropout):
        super().__init__()
        self.sa_block = _SelfAttention(latent_dim, num_head, dropout, time_embed_dim)
        self.ca_block = _CrossAttention(latent_dim, text_latent_dim, num_head, dropout, time_embed_dim)
        self.ffn = FFN(latent_dim, ffn_dim, dropout, time_embed_dim)


def _core_param_order(model):
    """(global table, per-layer tables) as lists of parameter groups in hig.h table order.
    A group with several parameters must be laid out contiguously (fused GEMM operand); an empty
    group is a NULL table entry (the two-person entries of a single-person model)."""
    blocks = list(model.temporal_decoder_blocks)

    def sty_blocks(b):  # stylization order the kernels index `ss` with: sa, ca, [int_ca], ffn
        ic = [b.int_ca_block.proj_out] if hasattr(b, "int_ca_block") else []
        return [b.sa_block.proj_out, b.ca_block.proj_out] + ic + [b.ffn.proj_out]

    stys = [s for b in blocks for s in sty_blocks(b)]
    two = hasattr(model, "joint_embed2")
    glob = [
        [model.sequence_embedding],
        [model.joint_embed.weight], [model.joint_embed.bias],
        [model.time_embed[0].weight], [model.time_embed[0].bias],
        [model.time_embed[2].weight], [model.time_embed[2].bias],
        [s.emb_layers[1].weight for s in stys], [s.emb_layers[1].bias for s in stys],
        [model.out.weight], [model.out.bias],
        [model.joint_embed2.weight] if two else [], [model.joint_embed2.bias] if two else [],
        [model.out2.weight] if two else [], [model.out2.bias] if two else [],
    ]
    layers = []
    for b in blocks:
        sa, ca, ff = b.sa_block, b.ca_block, b.ffn
        lay = [
            [sa.norm.weight], [sa.norm.bias],
            [sa.query.weight, sa.key.weight, sa.value.weight], [sa.query.bias, sa.key.bias, sa.value.bias],
            [sa.proj_out.norm.weight], [sa.proj_out.norm.bias],
            [sa.proj_out.out_layers[2].weight], [sa.proj_out.out_layers[2].bias],
            [ca.norm.weight], [ca.norm.bias], [ca.text_norm.weight], [ca.text_norm.bias],
            [ca.query.weight], [ca.query.bias],
            [ca.key.weight, ca.value.weight], [ca.key.bias, ca.value.bias],
            [ca.proj_out.norm.weight], [ca.proj_out.norm.bias],
            [ca.proj_out.out_layers[2].weight], [ca.proj_out.out_layers[2].bias],
            [ff.linear1.weight], [ff.linear1.bias], [ff.linear2.weight], [ff.linear2.bias],
            [ff.proj_out.norm.weight], [ff.proj_out.norm.bias],
            [ff.proj_out.out_layers[2].weight], [ff.proj_out.out_layers[2].bias],
        ]
        if hasattr(b, "int_ca_block"):
            ic = b.int_ca_block
            lay += [
                [ic.norm.weight], [ic.norm.bias],
                [ic.query.weight, ic.key.weight, ic.value.weight], [ic.query.bias, ic.key.bias, ic.value.bias],
                [ic.proj_out.norm.weight], [ic.proj_out.norm.bias],
                [ic.proj_out.out_layers[2].weight], [ic.proj_out.out_layers[2].bias],
            ]
        else:
            lay += [[] for _ in range(8)]
        layers.append(lay)
    assert len(glob) == _lib.NGLOBAL and all(len(l) == _lib.NLAYER for l in layers)
    return glob, layers


class _FlatParams:
    """One contiguous fp32 device buffer holding every denoiser-core parameter (and a twin for
    gradients) so that (a) q/k/v, key/value and the 3L stylization `emb_layers` matrices are
    single fused-GEMM operands, (b) the RCCL all-reduce and the fused clip+Adam see ONE buffer.
    The nn.Parameters become views into it; state-dict names/shapes are untouched.
    The text-head parameters (when the model has a HIP text head) follow the core in the same
    buffer -- `[0, core_numel)` core, `[core_numel, numel)` text head -- so the fused training step
    can update everything that is trainable with one all-reduce and one clip+Adam."""

    ALIGN = 64  # floats (256 B)

    def __init__(self, model):
        glob, layers = _core_param_order(model)
        groups = glob + [g for l in layers for g in l]
        self.params = [p for g in groups for p in g]          # denoiser core, hig.h table order
        dev = self.params[0].device
        assert all(p.device == dev and p.dtype == torch.float32 for p in self.params)
        offs, o = [], 0
        self.group_offsets = []
        for g in groups:
            o = (o + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            self.group_offsets.append(o if g else None)
            for p in g:
                offs.append(o)
                o += p.numel()
        self.core_numel = (o + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.offsets = offs
        # text head (HIG_T_* / HIG_TL_* order), one aligned group per tensor
        self.text_params = list(model._text_params()) if model._has_hip_text_head() else []
        if not all(p.device == dev and p.dtype == torch.float32 for p in self.text_params):
            self.text_params = []
        self.text_offsets, o = [], self.core_numel
        for p in self.text_params:
            self.text_offsets.append(o)
            o += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = o
        self.flat = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, off in zip(self.params + self.text_params, offs + self.text_offsets):
                view = self.flat[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self.grad = None
        self.grad_views = None
        self._ptr_key = None
        self._ptable = None
        self._gtable = None

    def table(self, base_ptr):
        n = len(self.group_offsets)
        arr = (C.c_void_p * n)()
        for i, off in enumerate(self.group_offsets):
            arr[i] = None if off is None else base_ptr + 4 * off
        return arr

    def param_table(self):
        key = self.flat.data_ptr()
        if self._ptr_key != key:
            self._ptable, self._ptr_key = self.table(key), key
        return self._ptable

    def ensure_grad(self):
        if self.grad is None:
            self.grad = torch.zeros_like(self.flat)
            self.grad_views = [self.grad[o:o + p.numel()].view(p.shape)
                               for p, o in zip(self.params, self.offsets)]
            self._gtable = self.table(self.grad.data_ptr())
        return self._gtable

    def text_grad_table(self, slots):
        """Pointer table (HIG_T_* order, NULL where `slots` is False) into the flat GRADIENT buffer."""
        self.ensure_grad()
        arr = (C.c_void_p * len(slots))()
        it = iter(self.text_offsets)
        for i, present in enumerate(slots):
            arr[i] = self.grad.data_ptr() + 4 * next(it) if present else None
        return arr

    def layer_buckets(self, num_layers, nsty, d, E):
        """Element ranges [(start, stop), ...] of the flat gradient that are FINAL when decoder layer l's backward is
        done (`per_layer[l]`: the layer's own parameters + its rows of the stacked stylization emb_layers.1.weight), and
        the ranges that are only final at the end of the backward (`tail`: the global parameters)."""
        ng, nl = _lib.NGLOBAL, _lib.NLAYER
        offs = self.group_offsets

        def first_of_layer(l):
            return next(o for o in offs[ng + l * nl: ng + (l + 1) * nl] if o is not None)

        starts = [first_of_layer(l) for l in range(num_layers)] + [self.core_numel]
        o_w = offs[7]                                    # HIG_P_STY_EMB_W
        rows = nsty * 2 * d * E
        per_layer = [[(starts[l], starts[l + 1]), (o_w + l * rows, o_w + (l + 1) * rows)] for l in range(num_layers)]
        tail = [(0, o_w), (o_w + num_layers * rows, starts[0])]
        return per_layer, tail

    def shadow16(self, version):
        """bf16 shadow of the flat buffer (same offsets, 2 bytes per element) and its pointer table, rebuilt by one
        cast kernel whenever `version` (the caller's view of "the parameters changed") moves."""
        if getattr(self, "_shadow", None) is None or self._shadow.data_ptr() != getattr(self, "_shadow_ptr", None):
            self._shadow = torch.empty(self.core_numel, device=self.flat.device, dtype=torch.bfloat16)
            self._shadow_ptr = self._shadow.data_ptr()
            n = len(self.group_offsets)
            self._stable = (C.c_void_p * n)()
            for i, off in enumerate(self.group_offsets):
                self._stable[i] = None if off is None else self._shadow_ptr + 2 * off
            self._shadow_version = None
        if self._shadow_version != (version, self.flat.data_ptr()):
            _lib.check(_lib.lib().hig_cast_bf16(_lib.ptr(self.flat), _lib.ptr(self._shadow), self.core_numel,
                                                _lib.stream_ptr()))
            self._shadow_version = (version, self.flat.data_ptr())
        return self._stable

    def shadow16_buffer(self):
        """The bf16 shadow tensor itself (allocated, contents unspecified): hig_clip_adam_shadow writes it."""
        if getattr(self, "_shadow", None) is None or self._shadow.data_ptr() != getattr(self, "_shadow_ptr", None):
            self.shadow16(object())       # allocate + one cast; the sentinel version never matches again
        return self._shadow

    def shadow16_mark_current(self, version):
        """The optimizer kernel has just written bf16(parameters) into the shadow: `version` is current without a cast pass."""
        self._shadow_version = (version, self.flat.data_ptr())

    def valid(self):
        base, end = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.numel
        return all(p.is_cuda and base <= p.data_ptr() < end for p in self.params[:3] + self.params[-3:]) \
            and all(p.data_ptr() == base + 4 * o for p, o in zip(self.params[:8], self.offsets[:8])) \
            and all(p.data_ptr() == base + 4 * o for p, o in zip(self.text_params, self.text_offsets))


_POISON = __import__("os").environ.get("HIG_POISON", "0") == "1"


class _WorkspacePool:
    """Scratch buffers keyed by (kind, bytes).  A forward that needs its activations kept for
    backward holds its buffer until backward returns it."""

    def __init__(self):
        self.free = {}

    def take(self, kind, nbytes, device):
        lst = self.free.setdefault((kind, nbytes, str(device)), [])
        buf = lst.pop() if lst else torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
        if _POISON:   # debugging aid (HIG_POISON=1): every scratch byte starts as NaN, so a kernel that reads
            buf.fill_(0xFF)   # workspace it did not write first shows up as NaN in the outputs
        return buf

    def give(self, kind, buf, device):
        self.free.setdefault((kind, buf.numel(), str(device)), []).append(buf)


class _DenoiserFn(torch.autograd.Function):
    """Autograd boundary: forward = hig_text_context + hig_denoiser_fwd(training=1),
    backward = hig_denoiser_bwd.  Parameter gradients come back as copies of the views of the
    flat gradient buffer (the fused trainer path reads the flat buffer directly instead)."""

    @staticmethod
    def forward(ctx, model, x, t, length, xf_proj, xf_out, *params):
        out, saved = model._launch_forward(x, t, length, xf_proj, xf_out, training=True)
        ctx.model, ctx.saved = model, saved
        ctx.save_for_backward(x, t, length, xf_out)
        ctx.need = (x.requires_grad, xf_proj.requires_grad, xf_out.requires_grad)
        return out

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        x, t, length, xf_out = ctx.saved_tensors
        dx, dxp, dxo = model._launch_backward(x, t, length, xf_out, ctx.saved, dout.contiguous(),
                                              want_dx=ctx.need[0])
        ctx.saved = None
        pg = tuple(g.clone() for g in model._flat.grad_views)
        return (None, dx if ctx.need[0] else None, None, None, dxp if ctx.need[1] else None,
                dxo if ctx.need[2] else None) + pg


class _TextHeadFn(torch.autograd.Function):
    """Autograd boundary of the text head (hig_text_head_fwd / _bwd): CLIP features (B, N, W) and the
    EOT indices in, (xf_proj, xf_out) out; the parameters are passed so autograd routes their
    gradients (one flat scratch buffer per backward, returned as views)."""

    @staticmethod
    def forward(ctx, model, clip_out, eot, *params):
        xf_proj, xf_out, saved = model._launch_text_head(clip_out, eot, training=True)
        ctx.model, ctx.saved = model, saved
        ctx.save_for_backward(clip_out, eot, xf_out)
        ctx.need_clip = clip_out.requires_grad
        return xf_proj, xf_out

    @staticmethod
    def backward(ctx, dxf_proj, dxf_out):
        clip_out, eot, xf_out = ctx.saved_tensors
        grads, dclip = ctx.model._launch_text_head_backward(clip_out, eot, xf_out, ctx.saved, dxf_out, dxf_proj,
                                                           ctx.need_clip)
        ctx.saved = None
        return (None, dclip, None) + tuple(grads)


class MotionTransformer(nn.Module):
    """Drop-in for the reference class (transformer.py:288-426)."""

    def __init__(self, input_feats, num_frames=240, latent_dim=512, ff_size=1024, num_layers=8,
                 num_heads=8, dropout=0, activation="gelu", num_text_layers=4, text_latent_dim=256,
                 text_ff_size=2048, text_num_heads=4, no_clip=False, no_eff=False, **kargs):
        super().__init__()
        if dropout != 0:
            # every reference tool uses the constructor default p=0; the fused
            # kernels implement exactly that
            raise NotImplementedError("dropout != 0 is not supported by the fused denoiser")
        self.num_frames = num_frames
        self.latent_dim = latent_dim
        self.ff_size = ff_size
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.dropout = dropout
        self.activation = activation
        self.input_feats = input_feats
        self.time_embed_dim = latent_dim * 4
        self.text_latent_dim = text_latent_dim
        self.no_eff = no_eff
        self.sequence_embedding = nn.Parameter(torch.randn(num_frames, latent_dim))

        # Text side: CLIP + the parameter containers of the text head (transformer.py:318-340)
        self.clip, _ = clip.load('ViT-B/32', "cpu")
        if no_clip:
            self.clip.initialize_parameters()
        else:
            set_requires_grad(self.clip, False)
        if text_latent_dim != 512:
            self.text_pre_proj = nn.Linear(512, text_latent_dim)
        else:
            self.text_pre_proj = nn.Identity()
        layer = nn.TransformerEncoderLayer(d_model=text_latent_dim, nhead=text_num_heads,
                                           dim_feedforward=text_ff_size, dropout=dropout,
                                           activation=activation)
        self.textTransEncoder = nn.TransformerEncoder(layer, num_layers=num_text_layers,
                                                      enable_nested_tensor=False)
        self.text_ln = nn.LayerNorm(text_latent_dim)
        self.text_proj = nn.Sequential(nn.Linear(text_latent_dim, self.time_embed_dim))

        # Denoiser core (parameters only; the math is in libhig.so)
        self.joint_embed = nn.Linear(self.input_feats, self.latent_dim)
        self.time_embed = nn.Sequential(
            nn.Linear(self.latent_dim, self.time_embed_dim),
            nn.SiLU(),
            nn.Linear(self.time_embed_dim, self.time_embed_dim),
        )
        self.temporal_decoder_blocks = nn.ModuleList(
            _DecoderLayer(latent_dim, text_latent_dim, self.time_embed_dim, ff_size, num_heads, dropout)
            for _ in range(num_layers))
        self.out = zero_module(nn.Linear(self.latent_dim, self.input_feats))

        # product arithmetic of the forward GEMMs: "f32" (exact fp32 MFMA, default), "bf16x3"
        # (split-bf16, ~1e-5 relative, 16x-rate MFMA) or "bf16"; HIG_PREC overrides the default
        import os
        self.precision = kargs.get("precision", os.environ.get("HIG_PREC", "f32"))
        # "hip": the text head runs through hig_text_head_fwd/_bwd; "torch": stock PyTorch-ROCm ops
        self.text_head = kargs.get("text_head", os.environ.get("HIG_TEXT_HEAD", "hip"))
        # "f32": fp32 activations / weights (default).  "bf16": bf16 activations + a bf16 shadow of the weight
        # matrices, fp32 accumulation and statistics (BASELINE configs 3 / 5) -- inference only
        self.storage = kargs.get("storage", os.environ.get("HIG_STORAGE", "f32"))
        self._flat = None
        self._pool = _WorkspacePool()
        self._textctx_cache = None
        self._param_epoch = 0
        # inference only: keep the cross-attention text context of the last `xf_out` (it is step-invariant, so the
        # sampling loop computes it once); False = recompute on every forward, as the reference does (:144-150)
        self.cache_text_context = True

    # ---- nn.Module plumbing -------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._flat = None            # .to()/.cuda()/.float() re-home parameters: re-flatten lazily
        self._textctx_cache = None
        return r

    def params_changed(self):
        """Tell the model its parameters were modified behind autograd's back (a fused optimizer kernel writing the
        flat buffer, a checkpoint load): drops everything derived from them."""
        self._param_epoch += 1
        self._textctx_cache = None

    def _load_from_state_dict(self, *a, **k):
        self._textctx_cache = None
        return super()._load_from_state_dict(*a, **k)

    def _textctx_param_version(self):
        """Sum of the autograd version counters of every parameter the text context is built from.  The
        nn.Parameters are re-homed VIEWS of the flat buffer, so an in-place update through them (torch optimizers,
        load_state_dict, p.copy_) bumps THEIR counter, never the flat buffer's."""
        v = self._param_epoch
        for blk in self.temporal_decoder_blocks:
            ca = blk.ca_block
            for p in (ca.key.weight, ca.key.bias, ca.value.weight, ca.value.bias, ca.text_norm.weight, ca.text_norm.bias):
                v += p._version
        return v

    def _param_version(self):
        """Moves whenever any core parameter may have changed (see `_textctx_param_version`)."""
        return self._param_epoch + sum(p._version for p in self.flat_params().params)

    def _bf16(self):
        if self.storage not in ("f32", "bf16"):
            raise ValueError("storage must be 'f32' or 'bf16' (got %r)" % (self.storage,))
        return self.storage == "bf16"

    def flat_params(self):
        """Flat fp32 buffer aliasing every core parameter (built on first use on the device)."""
        if self._flat is None or not self._flat.valid():
            if not self.out.weight.is_cuda:
                raise RuntimeError("MotionTransformer: parameters must live on a ROCm device "
                                   "(call .to('cuda')); there is no CPU fallback")
            self._flat = _FlatParams(self)
        return self._flat

    def core_parameters(self):
        return list(self.flat_params().params)

    # ---- reference API ------------------------------------------------------------------
    def _clip_features(self, text, device):
        """Frozen CLIP text tower up to ln_final (transformer.py:381-388): tokens + (L, B, 512)."""
        with torch.no_grad():
            text = clip.tokenize(text, truncate=True).to(device)
            x = self.clip.token_embedding(text).type(self.clip.dtype)
            x = x + self.clip.positional_embedding.type(self.clip.dtype)
            x = x.permute(1, 0, 2)
            x = self.clip.transformer(x)
            x = self.clip.ln_final(x).type(self.clip.dtype)
        return text, x

    def _text_head_torch(self, text, x):
        """transformer.py:389-397 on stock torch ops (`text_head="torch"`)."""
        x = self.text_pre_proj(x)
        xf_out = self.textTransEncoder(x)
        xf_out = self.text_ln(xf_out)
        xf_proj = self.text_proj(xf_out[text.argmax(dim=-1), torch.arange(xf_out.shape[1])])
        xf_out = xf_out.permute(1, 0, 2)
        return xf_proj, xf_out

    def encode_text(self, text, device):
        """transformer.py:380-397.  CLIP stays on its own (frozen / stubbed) ops; the trainable head
        after it -- text_pre_proj, the post-norm encoder, text_ln, EOT gather, text_proj -- runs in
        the HIP kernels."""
        text, x = self._clip_features(text, device)
        return self._text_head(text, x)

    def _text_head(self, text, x):
        """tokens (B, N) + CLIP features (N, B, W) -> xf_proj (B, E), xf_out (B, N, Lt)."""
        if self.text_head == "torch":
            return self._text_head_torch(text, x)
        if not x.is_cuda:
            raise RuntimeError("encode_text: ROCm device required (no CPU fallback); pass text_head='torch' "
                               "to run the text head on stock PyTorch ops")
        clip_out = x.permute(1, 0, 2).float().contiguous()
        eot = text.argmax(dim=-1).contiguous()
        tp = self._text_params()
        if torch.is_grad_enabled() and (clip_out.requires_grad or any(p.requires_grad for p in tp)):
            return _TextHeadFn.apply(self, clip_out, eot, *tp)
        xf_proj, xf_out, _ = self._launch_text_head(clip_out, eot, training=False)
        return xf_proj, xf_out

    def generate_src_mask(self, T, length):
        """(B, T) float CPU mask, mask[i, j] = j < length[i] (transformer.py:399-405); the
        trainer calls this directly (ddpm_trainer.py:116-119).  The kernels build their own
        mask from `length` on the device."""
        length = torch.as_tensor(length).detach().to("cpu", torch.int64).view(-1)
        return (torch.arange(T)[None, :] < length[:, None]).float()

    def dims(self, B, T, N):
        return _lib.Dims(B=B, T=T, F=self.input_feats, d=self.latent_dim, H=self.num_heads,
                         ff=self.ff_size, L=self.num_layers, N=N, Lt=self.text_latent_dim,
                         num_frames=self.num_frames,
                         attn_kind=_lib.ATTN_FULL if self.no_eff else _lib.ATTN_LINEAR,
                         prec={"f32": _lib.PREC_F32, "bf16x3": _lib.PREC_BF16X3, "bf16": _lib.PREC_BF16}[self.precision],
                         storage=_lib.STORE_BF16 if getattr(self, "storage", "f32") == "bf16" else _lib.STORE_F32)

    def forward(self, x, timesteps, length=None, text=None, xf_proj=None, xf_out=None):
        """x: (B, T, F) -> (B, T, F)   (transformer.py:407-426)."""
        B, T = x.shape[0], x.shape[1]
        if xf_proj is None or xf_out is None:
            xf_proj, xf_out = self.encode_text(text, x.device)
        if not x.is_cuda:
            raise RuntimeError("MotionTransformer.forward: ROCm device tensors required "
                               "(no CPU fallback; the CPU restatement lives in oracle/ for tests only)")
        assert x.shape[2] == self.input_feats and T <= self.num_frames
        return self._run(x, timesteps, length, xf_proj, xf_out)

    def _run(self, x, timesteps, length, xf_proj, xf_out):
        """Normalise the inputs and launch (through autograd when anything requires grad)."""
        B, T = x.shape[0], x.shape[1]
        dev = x.device
        x = x.float().contiguous()
        t = timesteps.to(dev).long().contiguous()
        assert t.shape == (B,)
        if length is None:
            length = torch.full((B,), T, dtype=torch.int64, device=dev)
        else:
            length = torch.as_tensor(length).to(dev).long().contiguous()
        xf_proj = xf_proj.float().contiguous()
        xf_out = xf_out.float().contiguous()
        fp = self.flat_params()
        needs_grad = torch.is_grad_enabled() and (
            x.requires_grad or xf_proj.requires_grad or xf_out.requires_grad
            or any(p.requires_grad for p in fp.params))
        if needs_grad:
            self._check_bf16_training()
            return _DenoiserFn.apply(self, x, t, length, xf_proj, xf_out, *fp.params)
        out, _ = self._launch_forward(x, t, length, xf_proj, xf_out, training=False)
        return out

    def _check_bf16_training(self):
        """storage='bf16' trains the single-person AND the two-person model with linear attention (hig_denoiser_fwd_bf16_train /
        _bwd_bf16: fp32 master weights, gradients and optimizer state; bf16 activations and matrix products)."""
        if self._bf16() and self.no_eff:
            raise NotImplementedError("storage='bf16' trains with linear attention only (no_eff: train with fp32 storage; "
                                      "precision='bf16x3' / 'bf16' select reduced-precision products there)")

    # ---- launches -----------------------------------------------------------------------
    def _text_context(self, dims, xf_out, training):
        """Cross-attention text side (all layers).  Cached on (storage, version) of xf_out and the
        parameter version, so the 1000-step sampling loop computes it once (it is step-invariant)."""
        fp = self.flat_params()
        use_cache = not training and self.cache_text_context
        if use_cache:
            key = (xf_out.data_ptr(), xf_out._version, tuple(xf_out.shape), fp.flat.data_ptr(),
                   fp.flat._version, self._textctx_param_version(), self.precision, self.storage)
            if self._textctx_cache is not None and self._textctx_cache[0] == key:
                return self._textctx_cache[1]
        L = _lib.lib()
        nbytes = L.hig_textctx_bytes(C.byref(dims), int(training))
        if nbytes < 0:
            raise RuntimeError("libhig: " + _lib.last_error())
        if training:
            buf = self._pool.take("textctx_t", nbytes, xf_out.device)
        elif use_cache:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=xf_out.device)
        else:   # one reusable buffer: the caller's forward consumes it on the same stream before the next call
            buf = self._pool.take("textctx_i", nbytes, xf_out.device)
            self._pool.give("textctx_i", buf, xf_out.device)
        if self._bf16() and training:
            _lib.check(L.hig_text_context_bf16_train(C.byref(dims), fp.param_table(), fp.shadow16(self._param_version()),
                                                     _lib.ptr(xf_out), _lib.ptr(buf), _lib.stream_ptr()))
        elif self._bf16():
            _lib.check(L.hig_text_context_bf16(C.byref(dims), fp.param_table(), fp.shadow16(self._param_version()),
                                               _lib.ptr(xf_out), _lib.ptr(buf), _lib.stream_ptr()))
        else:
            _lib.check(L.hig_text_context(C.byref(dims), fp.param_table(), _lib.ptr(xf_out), _lib.ptr(buf),
                                          int(training), _lib.stream_ptr()))
        if use_cache:
            self._textctx_cache = (key, buf, xf_out)  # keep xf_out alive so the key stays unique
        return buf

    @staticmethod
    def _frag16(W16):
        """(J, R) bf16 weight -> the same elements in the operand order of v_mfma_f32_32x32x16_bf16 (hig_weight_frag16):
        blocks of 32 rows x 16 reduce elements, each 64 lanes x 8 elements, lane = j % 32 + 32 ((r % 16) / 8)."""
        J, R = W16.shape
        return W16.view(J // 32, 32, R // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()

    def _derived16(self, fp):
        """Operands of the bf16-storage forward derived from the parameters, kept next to the bf16 shadow and rebuilt
        when the parameters change (`derived` of hig_denoiser_fwd_bf16): 13 L + 5 device pointers, NULL where a piece does
        not apply (the library then runs its LayerNorm kernel / pads per call).
        [13 l + 3 k + 0 .. 2] (d = 512 or 1024): the LayerNorm-folded projection k of layer l -- k = 0 self-attention q/k/v, 1
        cross-attention query, 2 q/k/v of the person <-> person attention (two-person model) -- as [W' (bf16), colsum, bias']
        with W' = gamma (.) W, colsum[j] = sum_r float(W'[j][r]), bias' = b + W beta: LayerNorm(x) W^T + b == rstd (x W'^T) -
        rstd mean colsum + bias' (transformer.py:108-110,144; interaction_transformer.py:181-190); applied wherever the
        producer of x wrote its row statistics.
        [13 l + 9 + s] (d = 512): the stylization-out weight of the self- (s = 0) / cross- (1) / person <-> person (2) attention /
        FFN (3) block in matrix-core operand order (_frag16) for the fused kernels hig_attn_out16 / hig_rows_out16.
        [13 L]: joint_embed weight (transformer.py:418) rounded to bf16 and padded to a multiple of 32 columns.
        [13 L + 1 .. 13 L + 4] (linear attention): the text side's key/value weights of ALL layers with their text_norm folded
        in (gamma (.) W, every layer's key rows then every layer's value rows, bf16), the bias' b + W beta in the same order
        (fp32), Lt ones, Lt zeros -- the per-call text side then runs ONE GEMM and ONE context build (transformer.py:146-152)."""
        ver = (self._param_version(), fp.flat.data_ptr())
        if getattr(self, "_derived", None) is None or self._derived[0] != ver:
            d, nl, ng, offs, L = self.latent_dim, _lib.NLAYER, _lib.NGLOBAL, fp.group_offsets, self.num_layers
            arr, bufs = (C.c_void_p * (13 * L + 5))(), []
            Lt = self.text_latent_dim
            wt, bt = [], []
            with torch.no_grad():
                for l in range(0 if self.no_eff else L):
                    def tgrp(idx, n):
                        o = offs[ng + l * nl + idx]
                        return fp.flat[o:o + n]
                    # text side, batched (see _derived32): gamma (.) [Wk; Wv] rounded to bf16, bias' = b + W beta in fp32
                    g_t, b_t = tgrp(10, Lt).double(), tgrp(11, Lt).double()
                    Wkv, bkv = tgrp(14, 2 * d * Lt).view(2 * d, Lt).double(), tgrp(15, 2 * d).double()
                    wt.append((Wkv * g_t[None, :]).float())
                    bt.append((bkv + Wkv @ b_t).float())
                if wt and Lt % 8 == 0:
                    Wt = torch.cat([w[:d] for w in wt] + [w[d:] for w in wt], 0).to(torch.bfloat16).contiguous()
                    Bt = torch.cat([b[:d] for b in bt] + [b[d:] for b in bt], 0).contiguous()
                    ones, zeros = torch.ones(Lt, device=Wt.device), torch.zeros(Lt, device=Wt.device)
                    bufs += [Wt, Bt, ones, zeros]
                    arr[13 * L + 1], arr[13 * L + 2], arr[13 * L + 3], arr[13 * L + 4] = Wt.data_ptr(), Bt.data_ptr(), ones.data_ptr(), zeros.data_ptr()
                for l in range(L if d in (512, 1024) else 0):
                    def grp(idx, n):
                        o = offs[ng + l * nl + idx]
                        return None if o is None else fp.flat[o:o + n]
                    # (norm weight, norm bias, Linear weight, Linear bias, output rows) by index in the layer table (hig.h)
                    for k, (nw, nb, wi, bi, rows) in enumerate(((0, 1, 2, 3, 3 * d), (8, 9, 12, 13, d), (28, 29, 30, 31, 3 * d))):
                        gamma = grp(nw, d)
                        if gamma is None:           # (the interaction block exists in the two-person model only)
                            continue
                        beta, W, b = grp(nb, d), grp(wi, rows * d).view(rows, d), grp(bi, rows)
                        Wp = (W * gamma[None, :]).to(torch.bfloat16).contiguous()
                        cs = Wp.float().sum(dim=1).contiguous()
                        bp = (b + W @ beta).contiguous()
                        bufs += [Wp, cs, bp]
                        arr[13 * l + 3 * k], arr[13 * l + 3 * k + 1], arr[13 * l + 3 * k + 2] = Wp.data_ptr(), cs.data_ptr(), bp.data_ptr()
                    # stylization-out weights of the attention blocks in matrix-core operand order (hig_attn_out16: d = 512)
                    for s, wi in enumerate((6, 18, 34, 26)):
                        w = grp(wi, d * d) if d == 512 else None
                        if w is not None:
                            wf = self._frag16(w.view(d, d).to(torch.bfloat16))
                            bufs.append(wf)
                            arr[13 * l + 9 + s] = wf.data_ptr()
                F = self.input_feats
                Fp = (F + 31) // 32 * 32
                wj = torch.zeros(d, Fp, device=fp.flat.device, dtype=torch.bfloat16)
                wj[:, :F] = self.joint_embed.weight.detach().to(torch.bfloat16)
                bufs.append(wj)
                arr[13 * L] = wj.data_ptr()
            self._derived = (ver, arr, bufs)
        return self._derived[1]

    def _derived32(self, fp):
        """Operands of the fp32 inference forward derived from the parameters, rebuilt when they change (`derived32` of
        hig_denoiser_fwd_x): per layer the LayerNorm-folded projections k = 0 (self-attention q/k/v) and k = 1 (cross-attention
        query) as [W' = gamma (.) W, colsum = row sums of W', bias' = b + W beta] -- LayerNorm(x) W^T + b == rstd (x W'^T) - rstd
        mean colsum + bias' (transformer.py:108-110,144), applied where the producer of x wrote its row statistics; and
        [6 L .. 6 L + 3] the text side's key/value weights of ALL layers with their text_norm folded in, stacked (L 2d, Lt: every
        layer's key rows, then every layer's value rows), the
        stacked bias', and a ones / zeros vector (the affine-free LayerNorm of the text rows) -- the per-call text side then runs
        one GEMM instead of L (transformer.py:146,150).  fp64 arithmetic for the derived vectors, fp32 storage; latent_dim % 128
        == 0 only (NULL table otherwise)."""
        d = self.latent_dim
        if d % 128 != 0 or d > 1024:
            return None
        ver = (self._param_version(), fp.flat.data_ptr())
        if getattr(self, "_derived_f32", None) is None or self._derived_f32[0] != ver:
            nl, ng, offs, L = _lib.NLAYER, _lib.NGLOBAL, fp.group_offsets, self.num_layers
            arr, bufs = (C.c_void_p * (6 * L + 4))(), []
            Lt = self.text_latent_dim
            wt, bt = [], []
            with torch.no_grad():
                for l in range(L):
                    def grp(idx, n):
                        o = offs[ng + l * nl + idx]
                        return fp.flat[o:o + n]
                    # text side (transformer.py:146,150): [key; value](LN_text(xf)) = xhat (gamma (.) W)^T + (W beta + b) with
                    # xhat = (xf - mean) rstd the same in every layer -- the folded weights of ALL layers stacked, one GEMM
                    g_t, b_t = grp(10, Lt).double(), grp(11, Lt).double()
                    Wkv, bkv = grp(14, 2 * d * Lt).view(2 * d, Lt).double(), grp(15, 2 * d).double()
                    wt.append((Wkv * g_t[None, :]).float())         # rows [0, d): key, [d, 2 d): value
                    bt.append((bkv + Wkv @ b_t).float())
                    for k, (nw, nb, wi, bi, rows) in enumerate(((0, 1, 2, 3, 3 * d), (8, 9, 12, 13, d))):
                        gamma, beta = grp(nw, d).double(), grp(nb, d).double()
                        W, b = grp(wi, rows * d).view(rows, d).double(), grp(bi, rows).double()
                        Wp = (W * gamma[None, :]).float().contiguous()
                        cs = Wp.double().sum(dim=1).float().contiguous()
                        bp = (b + W @ beta).float().contiguous()
                        bufs += [Wp, cs, bp]
                        arr[6 * l + 3 * k], arr[6 * l + 3 * k + 1], arr[6 * l + 3 * k + 2] = Wp.data_ptr(), cs.data_ptr(), bp.data_ptr()
                if not self.no_eff:
                    # all keys in front of all values: [K_0 .. K_{L-1} | V_0 .. V_{L-1}] (one context-build launch for all layers)
                    Wt = torch.cat([w[:d] for w in wt] + [w[d:] for w in wt], 0).contiguous()
                    Bt = torch.cat([b[:d] for b in bt] + [b[d:] for b in bt], 0).contiguous()
                    ones, zeros = torch.ones(Lt, device=Wt.device), torch.zeros(Lt, device=Wt.device)
                    bufs += [Wt, Bt, ones, zeros]
                    arr[6 * L], arr[6 * L + 1], arr[6 * L + 2], arr[6 * L + 3] = Wt.data_ptr(), Bt.data_ptr(), ones.data_ptr(), zeros.data_ptr()
            self._derived_f32 = (ver, arr, bufs)
        return self._derived_f32[1]

    def _launch_forward(self, x, t, length, xf_proj, xf_out, training):
        B, T, N = x.shape[0], x.shape[1], xf_out.shape[1]
        assert xf_out.shape == (B, N, self.text_latent_dim) and xf_proj.shape == (B, self.time_embed_dim)
        L = _lib.lib()
        fp = self.flat_params()
        dims = self.dims(B, T, N)
        if training:
            self._check_bf16_training()
        nbytes = L.hig_workspace_bytes(C.byref(dims), int(training))
        if nbytes < 0:
            raise RuntimeError("libhig: " + _lib.last_error())
        out = torch.empty(B, T, self.input_feats, device=x.device, dtype=torch.float32)
        if not training and not self._bf16() and not self.cache_text_context:
            # the reference's per-call forward (text side recomputed every call, transformer.py:144-150) as ONE library call:
            # the text side runs on a side stream next to the first layers (hig_denoiser_fwd_text)
            tbytes = L.hig_textctx_bytes(C.byref(dims), 0)
            if tbytes < 0:
                raise RuntimeError("libhig: " + _lib.last_error())
            textctx = self._pool.take("textctx_i", tbytes, x.device)
            ws = self._pool.take("fwd_i", nbytes, x.device)
            _lib.check(L.hig_denoiser_fwd_x(C.byref(dims), fp.param_table(), self._derived32(fp), _lib.ptr(x), _lib.ptr(t),
                                            _lib.ptr(length), _lib.ptr(xf_proj), _lib.ptr(xf_out), _lib.ptr(textctx), _lib.ptr(out),
                                            _lib.ptr(ws), 0, _lib.stream_ptr()))
            self._pool.give("fwd_i", ws, x.device)
            self._pool.give("textctx_i", textctx, x.device)
            return out, None
        if not training and self._bf16() and not self.cache_text_context:
            # same for bf16 storage (hig_denoiser_fwd_bf16_x): embedding chain and text side next to the frame-row launches
            tbytes = L.hig_textctx_bytes(C.byref(dims), 0)
            if tbytes < 0:
                raise RuntimeError("libhig: " + _lib.last_error())
            textctx = self._pool.take("textctx_i", tbytes, x.device)
            ws = self._pool.take("fwd_i", nbytes, x.device)
            _lib.check(L.hig_denoiser_fwd_bf16_x(C.byref(dims), fp.param_table(), fp.shadow16(self._param_version()),
                                                 self._derived16(fp), _lib.ptr(x), _lib.ptr(t), _lib.ptr(length), _lib.ptr(xf_proj),
                                                 _lib.ptr(xf_out), _lib.ptr(textctx), _lib.ptr(out), _lib.ptr(ws), _lib.stream_ptr()))
            self._pool.give("fwd_i", ws, x.device)
            self._pool.give("textctx_i", textctx, x.device)
            return out, None
        textctx = self._text_context(dims, xf_out, training)
        ws = self._pool.take("fwd_t" if training else "fwd_i", nbytes, x.device)
        if self._bf16() and training:
            _lib.check(L.hig_denoiser_fwd_bf16_train(C.byref(dims), fp.param_table(), fp.shadow16(self._param_version()),
                                                     _lib.ptr(x), _lib.ptr(t), _lib.ptr(length), _lib.ptr(xf_proj),
                                                     _lib.ptr(textctx), _lib.ptr(out), _lib.ptr(ws), _lib.stream_ptr()))
            return out, (dims, ws, textctx)
        if self._bf16():
            _lib.check(L.hig_denoiser_fwd_bf16_x(C.byref(dims), fp.param_table(), fp.shadow16(self._param_version()),
                                                 self._derived16(fp), _lib.ptr(x), _lib.ptr(t), _lib.ptr(length),
                                                 _lib.ptr(xf_proj), None, _lib.ptr(textctx), _lib.ptr(out), _lib.ptr(ws),
                                                 _lib.stream_ptr()))
            self._pool.give("fwd_i", ws, x.device)
            return out, None
        if not training:   # inference: the LayerNorm-folded projections (derived per parameter version)
            _lib.check(L.hig_denoiser_fwd_x(C.byref(dims), fp.param_table(), self._derived32(fp), _lib.ptr(x), _lib.ptr(t),
                                            _lib.ptr(length), _lib.ptr(xf_proj), None, _lib.ptr(textctx), _lib.ptr(out), _lib.ptr(ws),
                                            0, _lib.stream_ptr()))
            self._pool.give("fwd_i", ws, x.device)
            return out, None
        _lib.check(L.hig_denoiser_fwd(C.byref(dims), fp.param_table(), _lib.ptr(x), _lib.ptr(t),
                                      _lib.ptr(length), _lib.ptr(xf_proj), _lib.ptr(textctx), _lib.ptr(out),
                                      _lib.ptr(ws), int(training), _lib.stream_ptr()))
        return out, (dims, ws, textctx)

    def _launch_backward(self, x, t, length, xf_out, saved, dout, want_dx=False, layer_hook=None, comm_stream=None):
        """layer_hook(l): called on the host once decoder layer l's backward is enqueued and `comm_stream` (a
        torch.cuda.Stream) has been made to wait for it -- the data-parallel exchange of layer l's gradients can start
        there while the earlier layers are still in backward (hig_denoiser_bwd_hooked)."""
        dims, ws, textctx = saved
        L = _lib.lib()
        fp = self.flat_params()
        gtable = fp.ensure_grad()
        B, T, N = dims.B, dims.T, dims.N
        dev = x.device
        nb = L.hig_bwd_workspace_bytes(C.byref(dims))
        bws = self._pool.take("bwd", nb, dev)
        dx = torch.empty_like(x) if want_dx else None
        dxp = torch.empty(B, self.time_embed_dim, device=dev, dtype=torch.float32)
        dxo = torch.empty(B, N, self.text_latent_dim, device=dev, dtype=torch.float32)
        if dims.storage == _lib.STORE_BF16 and layer_hook is None:
            _lib.check(L.hig_denoiser_bwd_bf16(C.byref(dims), fp.param_table(), fp.shadow16(self._param_version()), _lib.ptr(x),
                                               _lib.ptr(t), _lib.ptr(length), _lib.ptr(xf_out), _lib.ptr(textctx), _lib.ptr(ws),
                                               _lib.ptr(dout), gtable, _lib.ptr(dx), _lib.ptr(dxp), _lib.ptr(dxo), _lib.ptr(bws),
                                               _lib.stream_ptr()))
        elif dims.storage == _lib.STORE_BF16:
            errors = []

            def _hook16(_user, layer):        # an exception must not unwind through the C frames
                try:
                    layer_hook(int(layer))
                except BaseException as e:  # noqa: BLE001
                    errors.append(e)

            cb = _lib.LAYER_HOOK(_hook16)
            _lib.check(L.hig_denoiser_bwd_bf16_hooked(
                C.byref(dims), fp.param_table(), fp.shadow16(self._param_version()), _lib.ptr(x), _lib.ptr(t), _lib.ptr(length),
                _lib.ptr(xf_out), _lib.ptr(textctx), _lib.ptr(ws), _lib.ptr(dout), gtable, _lib.ptr(dx), _lib.ptr(dxp), _lib.ptr(dxo),
                _lib.ptr(bws), _lib.stream_ptr(), cb, None, None if comm_stream is None else C.c_void_p(comm_stream.cuda_stream)))
            if errors:
                raise errors[0]
        elif layer_hook is None:
            _lib.check(L.hig_denoiser_bwd(C.byref(dims), fp.param_table(), _lib.ptr(x), _lib.ptr(t),
                                          _lib.ptr(length), _lib.ptr(xf_out), _lib.ptr(textctx), _lib.ptr(ws),
                                          _lib.ptr(dout), gtable, _lib.ptr(dx), _lib.ptr(dxp), _lib.ptr(dxo),
                                          _lib.ptr(bws), _lib.stream_ptr()))
        else:
            errors = []

            def _hook(_user, layer):          # an exception must not unwind through the C frames
                try:
                    layer_hook(int(layer))
                except BaseException as e:  # noqa: BLE001
                    errors.append(e)

            cb = _lib.LAYER_HOOK(_hook)
            _lib.check(L.hig_denoiser_bwd_hooked(
                C.byref(dims), fp.param_table(), _lib.ptr(x), _lib.ptr(t), _lib.ptr(length), _lib.ptr(xf_out),
                _lib.ptr(textctx), _lib.ptr(ws), _lib.ptr(dout), gtable, _lib.ptr(dx), _lib.ptr(dxp), _lib.ptr(dxo),
                _lib.ptr(bws), _lib.stream_ptr(), cb, None,
                None if comm_stream is None else C.c_void_p(comm_stream.cuda_stream)))
            if errors:
                raise errors[0]
        self._pool.give("bwd", bws, dev)
        self._pool.give("fwd_t", ws, dev)
        self._pool.give("textctx_t", textctx, dev)
        return dx, dxp, dxo


    # ---- text head (hig_text_head_*) -------------------------------------------------------
    def _has_hip_text_head(self):
        """True when this model owns a text head the HIP kernels cover (not cap_id, head dim in 8..128)."""
        if getattr(self, "text_head", "hip") != "hip" or not hasattr(self, "textTransEncoder"):
            return False
        l0 = self.textTransEncoder.layers[0]
        return self.text_latent_dim // l0.self_attn.num_heads in (8, 16, 32, 64, 128)

    def _text_params(self):
        """Text-head parameters in hig.h table order (HIG_T_*, then HIG_TL_* per layer)."""
        pre = self.text_pre_proj
        glob = [getattr(pre, "weight", None), getattr(pre, "bias", None), self.text_ln.weight, self.text_ln.bias,
                self.text_proj[0].weight, self.text_proj[0].bias]
        lay = []
        for l in self.textTransEncoder.layers:
            lay += [l.self_attn.in_proj_weight, l.self_attn.in_proj_bias, l.self_attn.out_proj.weight,
                    l.self_attn.out_proj.bias, l.norm1.weight, l.norm1.bias, l.linear1.weight, l.linear1.bias,
                    l.linear2.weight, l.linear2.bias, l.norm2.weight, l.norm2.bias]
        assert len(glob) == _lib.T_NGLOBAL and len(lay) == _lib.T_NLAYER * len(self.textTransEncoder.layers)
        self._text_slots = [p is not None for p in glob + lay]
        return [p for p in glob + lay if p is not None]

    def _text_dims(self, B, N, W):
        l0 = self.textTransEncoder.layers[0]
        return _lib.TextDims(B=B, N=N, W=W, Lt=self.text_latent_dim, H=l0.self_attn.num_heads,
                             ff=l0.linear1.out_features, L=len(self.textTransEncoder.layers), E=self.time_embed_dim,
                             prec={"f32": _lib.PREC_F32, "bf16x3": _lib.PREC_BF16X3, "bf16": _lib.PREC_BF16}[self.precision])

    def _text_table(self, tensors):
        """c_void_p table with NULL in the slots of an Identity text_pre_proj."""
        arr = (C.c_void_p * len(self._text_slots))()
        it = iter(tensors)
        for i, present in enumerate(self._text_slots):
            arr[i] = next(it).data_ptr() if present else None
        return arr

    def _launch_text_head(self, clip_out, eot, training):
        B, N, W = clip_out.shape
        L = _lib.lib()
        tp = self._text_params()
        for p in tp:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError("text head: parameters must be contiguous fp32 ROCm tensors")
        dims = self._text_dims(B, N, W)
        nbytes = L.hig_text_head_workspace_bytes(C.byref(dims), int(training))
        if nbytes < 0:
            raise RuntimeError("libhig: %s (text_head='torch' selects stock PyTorch ops)" % _lib.last_error())
        dev = clip_out.device
        ws = self._pool.take("txt_t" if training else "txt_i", nbytes, dev)
        xf_out = torch.empty(B, N, self.text_latent_dim, device=dev, dtype=torch.float32)
        xf_proj = torch.empty(B, self.time_embed_dim, device=dev, dtype=torch.float32)
        _lib.check(L.hig_text_head_fwd(C.byref(dims), self._text_table(tp), _lib.ptr(clip_out), _lib.ptr(eot),
                                       _lib.ptr(xf_out), _lib.ptr(xf_proj), _lib.ptr(ws), int(training),
                                       _lib.stream_ptr()))
        if not training:
            self._pool.give("txt_i", ws, dev)
            return xf_proj, xf_out, None
        return xf_proj, xf_out, (dims, ws)

    def _launch_text_head_backward(self, clip_out, eot, xf_out, saved, dxf_out, dxf_proj, want_dclip,
                                   into_flat=False):
        """into_flat: write the parameter gradients into the flat gradient buffer (fused training step)
        instead of a fresh scratch buffer (autograd path)."""
        dims, ws = saved
        L = _lib.lib()
        tp = self._text_params()
        dev = clip_out.device
        if into_flat:
            fp = self.flat_params()
            assert len(fp.text_params) == len(tp), "text head is not part of the flat parameter buffer"
            grads, gtable = None, fp.text_grad_table(self._text_slots)
        else:
            offs, o = [], 0
            for p in tp:
                offs.append(o)
                o += (p.numel() + 63) // 64 * 64
            gflat = torch.empty(o, device=dev, dtype=torch.float32)
            grads = [gflat[off:off + p.numel()].view(p.shape) for p, off in zip(tp, offs)]
            gtable = self._text_table(grads)
        dclip = torch.empty_like(clip_out) if want_dclip else None
        bws = self._pool.take("txt_bwd", L.hig_text_head_bwd_workspace_bytes(C.byref(dims)), dev)
        _lib.check(L.hig_text_head_bwd(
            C.byref(dims), self._text_table(tp), _lib.ptr(clip_out), _lib.ptr(eot), _lib.ptr(xf_out), _lib.ptr(ws),
            _lib.ptr(None if dxf_out is None else dxf_out.contiguous()),
            _lib.ptr(None if dxf_proj is None else dxf_proj.contiguous()),
            gtable, _lib.ptr(dclip), _lib.ptr(bws), _lib.stream_ptr()))
        self._pool.give("txt_bwd", bws, dev)
        self._pool.give("txt_t", ws, dev)
        return grads, dclip
