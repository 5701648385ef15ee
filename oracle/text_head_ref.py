"""ORACLE (test infrastructure, never the product path).

CPU restatement of the trainable text head of `MotionTransformer.encode_text`
(reference: codes/models/transformer.py:324-340 construction, :389-397 use) over a {name: tensor}
dict with the reference's state-dict names.  The reference builds it from torch's
nn.TransformerEncoderLayer(d_model=Lt, nhead, dim_feedforward, dropout=0, activation="gelu")
(post-norm, seq-first); restated here batch-first with explicit matmuls:
    x = norm1(x + out_proj(softmax(q k^T / sqrt(hd)) v));  x = norm2(x + linear2(gelu(linear1(x))))
Pinned by tests/golden/g10_text_head.npz (outputs + gradients of the reference's own encode_text).
"""
import math

import torch
import torch.nn.functional as F


def _ln(p, pre, x):
    return F.layer_norm(x, (x.shape[-1],), p[pre + ".weight"], p[pre + ".bias"], 1e-5)


def encoder_layer(p, pre, x, H):
    B, N, Lt = x.shape
    hd = Lt // H
    qkv = F.linear(x, p[pre + ".self_attn.in_proj_weight"], p[pre + ".self_attn.in_proj_bias"])
    q, k, v = (t.view(B, N, H, hd).transpose(1, 2) for t in qkv.split(Lt, dim=-1))     # (B, H, N, hd)
    w = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), dim=-1)
    att = (w @ v).transpose(1, 2).reshape(B, N, Lt)
    x = _ln(p, pre + ".norm1", x + F.linear(att, p[pre + ".self_attn.out_proj.weight"],
                                            p[pre + ".self_attn.out_proj.bias"]))
    f = F.linear(F.gelu(F.linear(x, p[pre + ".linear1.weight"], p[pre + ".linear1.bias"])),
                 p[pre + ".linear2.weight"], p[pre + ".linear2.bias"])
    return _ln(p, pre + ".norm2", x + f)


def text_head_forward(p, clip_out, eot, H, L):
    """clip_out (B, N, W) = CLIP ln_final output batch-first, eot (B,) = tokens.argmax(-1)
    -> xf_proj (B, E), xf_out (B, N, Lt)   (transformer.py:389-397)."""
    x = clip_out
    if "text_pre_proj.weight" in p:
        x = F.linear(x, p["text_pre_proj.weight"], p["text_pre_proj.bias"])
    for l in range(L):
        x = encoder_layer(p, "textTransEncoder.layers.%d" % l, x, H)
    xf_out = _ln(p, "text_ln", x)
    g = xf_out[torch.arange(x.shape[0]), eot]
    xf_proj = F.linear(g, p["text_proj.0.weight"], p["text_proj.0.bias"])
    return xf_proj, xf_out
