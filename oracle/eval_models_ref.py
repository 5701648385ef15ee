"""ORACLE (test infrastructure, never the product path).

CPU restatement of the reference's two evaluation classifiers as EvaluatorModelWrapper calls them
(codes/datasets/evaluator.py:479-493, under no_grad):
  * MotionEncoder.forward              codes/models/interaction_transformer.py:706-741
  * MotionConsistencyEvalModel.forward codes/models/interaction_transformer.py:803-829
over a {name: tensor} dict with the reference's state-dict names.  The reference runs torch's
nn.TransformerEncoder(batch_first=True, post-norm, gelu, dropout 0) with `src_key_padding_mask`;
restated with explicit matmuls (padded keys get -inf before the softmax).  Rows of padded tokens
differ from torch's nested-tensor fast path (which zeroes them) but never reach an output: the
MotionEncoder multiplies them by the mask, the consistency model reads the [cls] row only.
Pinned by tests/golden/g13_eval_models.npz (outputs of the reference's own modules).
"""
import math

import torch
import torch.nn.functional as F

from . import fill

# golden cases of tests/golden/g13 (T = tokens per person = frames + 1, F = dim_pose - 4)
EVAL_CASES = {
    "tiny": dict(B=3, T=12, F=15, d=64, H=8, ff=128, L=2, num_frames=16, length=[12, 7, 1]),
    "ntu": dict(B=2, T=91, F=259, d=128, H=4, ff=256, L=3, num_frames=196, length=[91, 40]),
}


def eval_inputs(cname, c):
    x1 = fill.tensor_for("g13.x1." + cname, (c["B"], c["T"], c["F"]))
    x2 = fill.tensor_for("g13.x2." + cname, (c["B"], c["T"], c["F"]))
    return x1, x2, torch.tensor(c["length"], dtype=torch.int64)


def _ln(p, pre, x):
    return F.layer_norm(x, (x.shape[-1],), p[pre + ".weight"], p[pre + ".bias"], 1e-5)


def encoder_layer(p, pre, x, H, kpad):
    """x (B, S, d); kpad (B, S) bool, True = not a key."""
    B, S, d = x.shape
    hd = d // H
    qkv = F.linear(x, p[pre + ".self_attn.in_proj_weight"], p[pre + ".self_attn.in_proj_bias"])
    q, k, v = (t.view(B, S, H, hd).transpose(1, 2) for t in qkv.split(d, dim=-1))
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    s = s.masked_fill(kpad[:, None, None, :], float("-inf"))
    att = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, S, d)
    x = _ln(p, pre + ".norm1", x + F.linear(att, p[pre + ".self_attn.out_proj.weight"],
                                            p[pre + ".self_attn.out_proj.bias"]))
    f = F.linear(F.gelu(F.linear(x, p[pre + ".linear1.weight"], p[pre + ".linear1.bias"])),
                 p[pre + ".linear2.weight"], p[pre + ".linear2.bias"])
    return _ln(p, pre + ".norm2", x + f)


def embed_person(p, x):
    """(B, T, F) -> (B, T, d): token 0 = joint_embed2 of the init-pose row's first 4 features,
    token t >= 1 = joint_embed1(x[t]) + sequence_embedding[t-1]   (:717-720)."""
    T = x.shape[1]
    move = F.linear(x[:, 1:], p["joint_embed1.weight"], p["joint_embed1.bias"]) + p["sequence_embedding"][None, :T - 1]
    init = F.linear(x[:, 0, :4], p["joint_embed2.weight"], p["joint_embed2.bias"])
    return torch.cat([init[:, None], move], dim=1)


def token_mask(T, length):
    """(B, T) float: 1 where t < length[b]   (generate_src_mask's per-person block, :695-704)."""
    return (torch.arange(T)[None, :] < torch.as_tensor(length).view(-1, 1)).float()


def num_layers(p):
    return 1 + max(int(k.split(".")[2]) for k in p if k.startswith("motionTransEncoder.layers."))


def motion_encoder_forward(p, x1, x2, length, H):
    """-> (logits (B, class_num), feature (B, d))."""
    B, T = x1.shape[:2]
    m = token_mask(T, length)
    mask = torch.cat([m, m], dim=1)
    h = torch.cat([embed_person(p, x1), embed_person(p, x2)], dim=1)
    for l in range(num_layers(p)):
        h = encoder_layer(p, "motionTransEncoder.layers.%d" % l, h, H, mask < 0.5)
    o = F.linear(h, p["out1.weight"], p["out1.bias"])
    for s in (0, T):
        o[:, s] = F.linear(h[:, s], p["out2.weight"], p["out2.bias"])
    feat = (o * mask[..., None]).sum(dim=1) / mask.sum(dim=1, keepdim=True)
    return F.linear(feat, p["fin_proj.0.weight"], p["fin_proj.0.bias"]), feat


def consistency_forward(p, x1, x2, length, H):
    """-> logits (B, class_num) read from the learned [cls] token."""
    B, T = x1.shape[:2]
    m = token_mask(T, length)
    mask = torch.cat([torch.ones(B, 1), m, m], dim=1)
    h = torch.cat([p["cls_input"].expand(B, 1, -1), embed_person(p, x1), embed_person(p, x2)], dim=1)
    for l in range(num_layers(p)):
        h = encoder_layer(p, "motionTransEncoder.layers.%d" % l, h, H, mask < 0.5)
    return F.linear(h[:, 0], p["cls_output.0.weight"], p["cls_output.0.bias"])


def param_shapes(kind, F_, d, ff, L, num_frames, class_num=None):
    """State-dict names / shapes of the reference modules (kind "enc" = MotionEncoder, "con" =
    MotionConsistencyEvalModel); tests/golden/g13 carries the reference's own key list to check this."""
    s = {"sequence_embedding": (num_frames, d), "init_pos_embedding": (1, d),
         "joint_embed1.weight": (d, F_), "joint_embed1.bias": (d,),
         "joint_embed2.weight": (d, 4), "joint_embed2.bias": (d,),
         "time_embed.0.weight": (4 * d, d), "time_embed.0.bias": (4 * d,),
         "time_embed.2.weight": (4 * d, 4 * d), "time_embed.2.bias": (4 * d,)}
    for l in range(L):
        pre = "motionTransEncoder.layers.%d." % l
        s.update({pre + "self_attn.in_proj_weight": (3 * d, d), pre + "self_attn.in_proj_bias": (3 * d,),
                  pre + "self_attn.out_proj.weight": (d, d), pre + "self_attn.out_proj.bias": (d,),
                  pre + "linear1.weight": (ff, d), pre + "linear1.bias": (ff,),
                  pre + "linear2.weight": (d, ff), pre + "linear2.bias": (d,),
                  pre + "norm1.weight": (d,), pre + "norm1.bias": (d,),
                  pre + "norm2.weight": (d,), pre + "norm2.bias": (d,)})
    if kind == "enc":
        c = 26 if class_num is None else class_num
        s.update({"out1.weight": (d, d), "out1.bias": (d,), "out2.weight": (d, d), "out2.bias": (d,),
                  "fin_proj.0.weight": (c, d), "fin_proj.0.bias": (c,)})
    else:
        c = 2 if class_num is None else class_num
        s.update({"cls_input": (1, 1, d), "cls_output.0.weight": (c, d), "cls_output.0.bias": (c,)})
    return s
