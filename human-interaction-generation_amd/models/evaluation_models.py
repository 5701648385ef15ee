"""MI355X-native forward of the reference's two evaluation classifiers:
`MotionEncoder` (codes/models/interaction_transformer.py:641-741) -- class logits plus the pooled
feature that FID / diversity / multimodality are computed on -- and `MotionConsistencyEvalModel`
(:743-829) -- real/fake logits from a learned [cls] token.  Same constructors, attributes,
state-dict keys (checkpoints `best_eval_model.pth` load with strict=True) and call signatures.

Inference only, which is how the reference's evaluation uses them (EvaluatorModelWrapper,
codes/datasets/evaluator.py:468-493: `.eval()`, under no_grad); training the classifiers
(tools/train_evaluation_model.py) is out of scope.  The whole forward is ONE C-ABI call,
`hig_eval_encoder_fwd` (include/hig.h, csrc/evalnet.hip); there is no CPU fallback.
"""
import ctypes as C
import os

import torch
from torch import nn

from .. import _lib
from .transformer import _WorkspacePool, zero_module

__all__ = ["MotionEncoder", "MotionConsistencyEvalModel"]

_PREC = {"f32": _lib.PREC_F32, "bf16x3": _lib.PREC_BF16X3, "bf16": _lib.PREC_BF16}


class _EvalEncoderBase(nn.Module):
    """Parameter containers shared by both classifiers (:654-690 / :760-790).  `time_embed` and
    `init_pos_embedding` are never used by the reference's forward but are part of its state dict."""

    _cls_token = 0

    def __init__(self, input_feats, num_frames, latent_dim, ff_size, num_layers, num_heads, dropout, activation,
                 kargs):
        super().__init__()
        if dropout != 0:
            raise NotImplementedError("dropout != 0 is not supported (the evaluator runs in eval mode)")
        if activation != "gelu":
            raise NotImplementedError("only the reference's activation='gelu' is built")
        self.num_frames = num_frames
        self.latent_dim = latent_dim
        self.ff_size = ff_size
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.dropout = dropout
        self.activation = activation
        self.input_feats = input_feats
        self.time_embed_dim = latent_dim * 4
        self.sequence_embedding = nn.Parameter(torch.randn(num_frames, latent_dim))
        self.init_pos_embedding = nn.Parameter(torch.randn(1, latent_dim))
        self.precision = kargs.get("precision", os.environ.get("HIG_PREC", "f32"))
        self._pool = _WorkspacePool()

    def _build_trunk(self):
        d = self.latent_dim
        self.joint_embed1 = nn.Linear(self.input_feats, d)
        self.joint_embed2 = nn.Linear(4, d)
        self.time_embed = nn.Sequential(nn.Linear(d, self.time_embed_dim), nn.SiLU(),
                                        nn.Linear(self.time_embed_dim, self.time_embed_dim))
        layer = nn.TransformerEncoderLayer(d_model=d, nhead=self.num_heads, dim_feedforward=self.ff_size,
                                           dropout=self.dropout, activation=self.activation, batch_first=True)
        self.motionTransEncoder = nn.TransformerEncoder(layer, num_layers=self.num_layers,
                                                        enable_nested_tensor=False)

    # hig.h table: HIG_EV_* globals, then per layer the HIG_TL_* block
    def _globals(self):
        raise NotImplementedError

    def _table(self):
        ps = list(self._globals())
        for layer in self.motionTransEncoder.layers:
            a = layer.self_attn
            ps += [a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                   layer.norm1.weight, layer.norm1.bias, layer.linear1.weight, layer.linear1.bias,
                   layer.linear2.weight, layer.linear2.bias, layer.norm2.weight, layer.norm2.bias]
        arr = (C.c_void_p * len(ps))()
        for i, p in enumerate(ps):
            if p is None:
                arr[i] = None
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError("evaluation classifier: parameters must be contiguous fp32 ROCm tensors "
                                   "(no CPU fallback; the CPU restatement lives in oracle/ for tests only)")
            arr[i] = p.data_ptr()
        return arr

    def _src_mask_block(self, T, length):
        length = torch.as_tensor(length).detach().to("cpu", torch.int64).view(-1)
        return (torch.arange(T)[None, :] < length[:, None]).float()

    def _launch(self, x1, x2, length, class_num, want_feature):
        if not (x1.is_cuda and x2.is_cuda):
            raise RuntimeError("%s.forward: ROCm device tensors required (no CPU fallback)" % type(self).__name__)
        if torch.is_grad_enabled() and self.training:
            raise NotImplementedError("%s: inference only -- call .eval() / torch.no_grad() like the reference's "
                                      "EvaluatorModelWrapper does" % type(self).__name__)
        B, T, F_ = x1.shape
        assert x2.shape == x1.shape and F_ == self.input_feats and 2 <= T <= self.num_frames + 1
        if length is None:
            raise ValueError("length is required (the reference indexes it unconditionally)")
        dev = x1.device
        x1 = x1.detach().float().contiguous()
        x2 = x2.detach().float().contiguous()
        length = torch.as_tensor(length).detach().to(dev, torch.int64).view(-1).contiguous()
        assert length.numel() == B
        dims = _lib.EvalDims(B=B, T=T, F=F_, d=self.latent_dim, H=self.num_heads, ff=self.ff_size,
                             L=self.num_layers, C=class_num, cls=self._cls_token, prec=_PREC[self.precision])
        L = _lib.lib()
        nbytes = L.hig_eval_encoder_workspace_bytes(C.byref(dims))
        if nbytes < 0:
            raise RuntimeError("libhig: " + _lib.last_error())
        ws = self._pool.take("eval_ws", nbytes, dev)
        logits = torch.empty(B, class_num, device=dev, dtype=torch.float32)
        feature = torch.empty(B, self.latent_dim, device=dev, dtype=torch.float32) if want_feature else None
        table = self._table()
        with torch.cuda.device(dev):
            _lib.check(L.hig_eval_encoder_fwd(C.byref(dims), table, _lib.ptr(x1), _lib.ptr(x2), _lib.ptr(length),
                                              _lib.ptr(logits), _lib.ptr(feature), _lib.ptr(ws), _lib.stream_ptr()))
        self._pool.give("eval_ws", ws, dev)   # stream-ordered reuse: the next call launches behind this one
        return logits, feature


class MotionEncoder(_EvalEncoderBase):
    """Drop-in for the reference class (interaction_transformer.py:641-741)."""

    def __init__(self, input_feats, num_frames=240, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8,
                 dropout=0, class_num=26, activation="gelu", **kargs):
        super().__init__(input_feats, num_frames, latent_dim, ff_size, num_layers, num_heads, dropout, activation,
                         kargs)
        self._build_trunk()
        self.out1 = zero_module(nn.Linear(latent_dim, latent_dim))
        self.out2 = zero_module(nn.Linear(latent_dim, latent_dim))
        self.fin_proj = nn.Sequential(nn.Linear(latent_dim, class_num))

    def _globals(self):
        return (self.sequence_embedding, self.joint_embed1.weight, self.joint_embed1.bias,
                self.joint_embed2.weight, self.joint_embed2.bias, self.out1.weight, self.out1.bias,
                self.out2.weight, self.out2.bias, self.fin_proj[0].weight, self.fin_proj[0].bias, None)

    def generate_src_mask(self, T, length):
        """(B, 2T) float CPU mask: each person's tokens t < length[b] (:695-704)."""
        m = self._src_mask_block(T, length)
        return torch.cat([m, m], dim=1)

    def forward(self, x1, x2, length=None, text=None, xf_proj=None, xf_out=None):
        """x1, x2: (B, T, input_feats) -> (class logits (B, class_num), pooled feature (B, latent_dim))."""
        return self._launch(x1, x2, length, self.fin_proj[0].out_features, True)


class MotionConsistencyEvalModel(_EvalEncoderBase):
    """Drop-in for the reference class (interaction_transformer.py:743-829)."""

    _cls_token = 1

    def __init__(self, input_feats, num_frames=240, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8,
                 dropout=0, interaction_class_num=26, class_num=2, activation="gelu", **kargs):
        super().__init__(input_feats, num_frames, latent_dim, ff_size, num_layers, num_heads, dropout, activation,
                         kargs)
        self.cls_input = nn.Parameter(torch.randn(1, 1, latent_dim))
        self._build_trunk()
        self.cls_output = nn.Sequential(nn.Linear(latent_dim, class_num))

    def _globals(self):
        return (self.sequence_embedding, self.joint_embed1.weight, self.joint_embed1.bias,
                self.joint_embed2.weight, self.joint_embed2.bias, None, None, None, None,
                self.cls_output[0].weight, self.cls_output[0].bias, self.cls_input)

    def generate_src_mask(self, T, length):
        """(B, 1 + 2T) float CPU mask: the [cls] token, then each person's tokens t < length[b] (:792-801)."""
        m = self._src_mask_block(T, length)
        return torch.cat([torch.ones(m.shape[0], 1), m, m], dim=1)

    def forward(self, x1, x2, length=None, text=None, xf_proj=None, xf_out=None):
        """x1, x2: (B, T, input_feats) -> logits (B, class_num)."""
        return self._launch(x1, x2, length, self.cls_output[0].out_features, False)[0]
