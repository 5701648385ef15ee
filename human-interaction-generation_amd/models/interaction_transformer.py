"""MI355X-native `MotionInteractionTransformer`: the two-person denoiser of the reference
(codes/models/interaction_transformer.py:396-616), same constructor, attributes, state-dict keys
and call signature, running through the same C ABI as the single-person model with
`hig_dims.two_person` set (include/hig.h).

Batch convention (reference :577-583): x = cat([person 1 of every pair, person 2 of every pair]);
token 0 of each sample is the init-pose row (4 features through `joint_embed2`, read back through
`out2`), tokens 1.. are motion frames.  Each decoder layer adds a person<->person linear cross
attention (`int_ca_block`, :167-207) between the text cross-attention and the FFN.

Only the linear-attention variant is built (`no_eff=True` raises: the reference's `no_eff` layer for
this model, :369-394, drops the interaction attention altogether and no reference tool selects it).
"""
import os

import torch
from torch import nn

from .. import _lib
from .transformer import (FFN, MotionTransformer, StylizationBlock, _CrossAttention, _SelfAttention,
                          _WorkspacePool, clip, set_requires_grad, timestep_embedding, zero_module)

__all__ = ["MotionInteractionTransformer", "timestep_embedding", "zero_module", "set_requires_grad"]


class _InteractionCrossAttention(nn.Module):
    """Parameters of LinearTemporalInteractionCrossAttention (:167-179): one LayerNorm shared by both
    persons, query / key / value (d, d), stylization projection."""

    def __init__(self, latent_dim, num_head, dropout, time_embed_dim):
        super().__init__()
        self.num_head = num_head
        self.norm = nn.LayerNorm(latent_dim)
        self.query = nn.Linear(latent_dim, latent_dim)
        self.key = nn.Linear(latent_dim, latent_dim)
        self.value = nn.Linear(latent_dim, latent_dim)
        self.dropout = nn.Dropout(dropout)
        self.proj_out = StylizationBlock(latent_dim, time_embed_dim, dropout)


class _InteractionDecoderLayer(nn.Module):
    """:334-367: sa_block -> ca_block -> [int_ca_block] -> ffn (submodule order = state-dict order)."""

    def __init__(self, latent_dim, text_latent_dim, time_embed_dim, ffn_dim, num_head, dropout, no_cross_attn):
        super().__init__()
        self.no_cross_attn = no_cross_attn
        self.sa_block = _SelfAttention(latent_dim, num_head, dropout, time_embed_dim)
        self.ca_block = _CrossAttention(latent_dim, text_latent_dim, num_head, dropout, time_embed_dim)
        if not no_cross_attn:
            self.int_ca_block = _InteractionCrossAttention(latent_dim, num_head, dropout, time_embed_dim)
        self.ffn = FFN(latent_dim, ffn_dim, dropout, time_embed_dim)


class MotionInteractionTransformer(MotionTransformer):
    """Drop-in for the reference class (interaction_transformer.py:396-616)."""

    def __init__(self, input_feats, num_frames=240, latent_dim=512, ff_size=1024, num_layers=8,
                 num_heads=8, dropout=0, activation="gelu", num_text_layers=4, text_latent_dim=256,
                 text_ff_size=2048, text_num_heads=4, no_clip=False, no_eff=False, no_cross_attn=False,
                 cap_id=False, **kargs):
        nn.Module.__init__(self)
        if dropout != 0:
            raise NotImplementedError("dropout != 0 is not supported by the fused denoiser")
        if no_eff:
            raise NotImplementedError("MotionInteractionTransformer: only the linear-attention layers are built")
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
        self.cap_id = cap_id
        self.no_eff = False
        self.no_cross_attn = no_cross_attn

        # Text side: CLIP / class embeddings + the parameter containers of the text head (:421-462)
        if self.cap_id:
            self.cap_embedding = nn.Parameter(torch.randn(43, text_latent_dim))
            self.text_proj = nn.Sequential(nn.Linear(text_latent_dim, self.time_embed_dim))
        else:
            self.clip, _ = clip.load('ViT-B/32', "cpu")
            self.no_clip = no_clip
            if no_clip:
                self.clip.d_type = self.clip.token_embedding.weight.dtype
                self.clip.initialize_parameters()
                for name in ("visual", "logit_scale", "text_projection"):
                    if hasattr(self.clip, name):
                        delattr(self.clip, name)
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

        # Denoiser core (parameters only; :464-509)
        self.sequence_embedding = nn.Parameter(torch.randn(num_frames, latent_dim))
        self.two_embed = True
        self.joint_embed = nn.Linear(self.input_feats, self.latent_dim)
        self.joint_embed2 = nn.Linear(4, self.latent_dim)
        self.time_embed = nn.Sequential(
            nn.Linear(self.latent_dim, self.time_embed_dim),
            nn.SiLU(),
            nn.Linear(self.time_embed_dim, self.time_embed_dim),
        )
        self.temporal_decoder_blocks = nn.ModuleList(
            _InteractionDecoderLayer(latent_dim, text_latent_dim, self.time_embed_dim, ff_size, num_heads,
                                     dropout, no_cross_attn)
            for _ in range(num_layers))
        self.out = zero_module(nn.Linear(self.latent_dim, self.input_feats))
        self.out2 = zero_module(nn.Linear(self.latent_dim, self.input_feats))

        self.precision = kargs.get("precision", os.environ.get("HIG_PREC", "f32"))
        self.text_head = kargs.get("text_head", os.environ.get("HIG_TEXT_HEAD", "hip"))
        self.storage = kargs.get("storage", os.environ.get("HIG_STORAGE", "f32"))   # "bf16": inference with bf16 activations
        self._flat = None
        self._pool = _WorkspacePool()
        self._textctx_cache = None
        self._param_epoch = 0
        self.cache_text_context = True

    # ---- reference API ------------------------------------------------------------------
    def load_my_state_dict(self, state_dict, opt):
        """Partial checkpoint loading (:511-530): language-only / motion-only / whatever matches."""
        own_state = self.state_dict()
        for name, param in state_dict.items():
            is_text = 'clip' in name or 'text' in name
            if opt.only_language:
                if name not in own_state or not is_text:
                    print(name)
                    continue
            elif opt.only_motion:
                if name not in own_state or is_text:
                    print(name)
                    continue
            else:
                if name not in own_state:
                    if not (opt.cap_id and is_text):
                        print(name)
                    continue
            if isinstance(param, torch.nn.parameter.Parameter):
                param = param.data
            own_state[name].copy_(param)
        self.params_changed()

    def encode_text(self, text, device):
        """:532-556; with `no_clip` the CLIP text tower is trained too (no no_grad)."""
        if not self.no_clip:
            return MotionTransformer.encode_text(self, text, device)
        text = clip.tokenize(text, truncate=True).to(device)
        x = self.clip.token_embedding(text).type(self.clip.d_type)
        x = x + self.clip.positional_embedding.type(self.clip.d_type)
        x = x.permute(1, 0, 2)
        x = self.clip.transformer(x)
        x = self.clip.ln_final(x).type(self.clip.d_type)
        return self._text_head(text, x)

    def get_class_embedding(self, text):
        """:558-563: caption ids -> learned class embeddings (cap_id models)."""
        text = torch.cat(text)
        xf_proj = self.cap_embedding[text]
        xf_out = xf_proj.unsqueeze(1)
        xf_proj = self.text_proj(xf_proj)
        return xf_proj, xf_out

    def dims(self, B, T, N):
        d = MotionTransformer.dims(self, B, T, N)
        d.two_person = 2 if self.no_cross_attn else 1
        return d

    def forward(self, x, timesteps, length=None, text=None, xf_proj=None, xf_out=None):
        """x: (2B, T, F) = cat([x1, x2]) -> (2B, T, F)   (:577-616)."""
        B2, T = x.shape[0], x.shape[1]
        if self.cap_id:
            xf_proj, xf_out = self.get_class_embedding(text)
        elif xf_proj is None or xf_out is None:
            xf_proj, xf_out = self.encode_text(text, x.device)
        if not x.is_cuda:
            raise RuntimeError("MotionInteractionTransformer.forward: ROCm device tensors required "
                               "(no CPU fallback; the CPU restatement lives in oracle/ for tests only)")
        assert B2 % 2 == 0 and x.shape[2] == self.input_feats and T - 1 <= self.num_frames
        return self._run(x, timesteps, length, xf_proj, xf_out)
