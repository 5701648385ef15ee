"""ORACLE (test infrastructure, never the product path).

CPU restatement of the two-person denoiser `MotionInteractionTransformer.forward`
(reference: codes/models/interaction_transformer.py:577-616 and the blocks it calls,
:167-207 interaction cross-attention, :334-367 decoder layer; linear-attention variant,
two_embed=True) over a {name: tensor} parameter dict with the reference's state-dict names.
Shares the single-person blocks with oracle/denoiser_ref.py (the reference's two files define
them identically).  Pinned by tests/golden/g8_interaction*.npz.
"""
import torch
import torch.nn.functional as F

from . import denoiser_ref as R


def interaction_cross_attention(p, pre, h, emb, mask, H):
    """LinearTemporalInteractionCrossAttention.forward (cross=True), :181-205.
    h = cat([x1, x2]) (2B, T, D); queries from the own stream, keys/values from the partner
    (the SAME norm is applied to both); key softmax masked, value NOT masked."""
    B2, T, D = h.shape
    B = B2 // 2
    partner = torch.cat([h[B:], h[:B]], dim=0)
    q = R._lin(p, pre + ".query", R._ln(p, pre + ".norm", h))
    pn = R._ln(p, pre + ".norm", partner)
    k = R._lin(p, pre + ".key", pn) + (1 - mask) * -1000000
    q = F.softmax(q.view(B2, T, H, -1), dim=-1)
    k = F.softmax(k.view(B2, T, H, -1), dim=1)
    v = R._lin(p, pre + ".value", pn).view(B2, T, H, -1)
    att = torch.einsum("bnhd,bnhl->bhdl", k, v)
    y = torch.einsum("bnhd,bhdl->bnhl", q, att).reshape(B2, T, D)
    return h + R.stylization(p, pre + ".proj_out", y, emb)


def interaction_forward(p, x, t, length, xf_proj, xf_out, num_heads, num_layers, no_cross_attn=False,
                        return_intermediates=False):
    """MotionInteractionTransformer.forward, :577-616.  x = cat([x1, x2]) (2B, T, F); token 0 of each
    sample is the init-pose row (first 4 features through joint_embed2, output through out2)."""
    B2, T = x.shape[:2]
    d = p["joint_embed.weight"].shape[0]
    te = R.timestep_embedding(t, d).to(x.dtype)
    emb = F.linear(F.silu(R._lin(p, "time_embed.0", te)), p["time_embed.2.weight"],
                   p["time_embed.2.bias"]) + xf_proj
    move = R._lin(p, "joint_embed", x[:, 1:]) + p["sequence_embedding"][None, :T - 1]
    init = R._lin(p, "joint_embed2", x[:, 0, :4])
    h = torch.cat([init[:, None], move], dim=1)
    mask = R.src_mask(T, length).to(x.dtype).unsqueeze(-1)
    inter = {"h0": h}
    for l in range(num_layers):
        pre = "temporal_decoder_blocks.%d" % l
        h = R.linear_self_attention(p, pre + ".sa_block", h, emb, mask, num_heads)
        h = R.linear_cross_attention(p, pre + ".ca_block", h, xf_out, emb, num_heads)
        if not no_cross_attn:
            h = interaction_cross_attention(p, pre + ".int_ca_block", h, emb, mask, num_heads)
            if l == 0:
                inter["int0"] = h
        h = R.ffn(p, pre + ".ffn", h, emb)
    out = torch.cat([R._lin(p, "out2", h[:, 0])[:, None], R._lin(p, "out", h[:, 1:])], dim=1).contiguous()
    if return_intermediates:
        return out, inter
    return out


def pit_loss(pred, target, mask):
    """DDPMMulTrainer.backward_G, PIT branch (mul_ddpm_trainer.py:234-243).  Batch layout
    [m1|c1, m1|c2, m2|c2, m2|c1] (4 groups of B pairs): per pair the cheaper of the two caption
    assignments, init-pose row scored on its first 4 features only."""
    l0 = ((pred[:, 0, :4] - target[:, 0, :4]) ** 2).mean(-1)
    l1 = ((pred[:, 1:] - target[:, 1:]) ** 2).mean(-1)
    l = torch.cat([l0[:, None], l1], dim=1)
    n = l.shape[0]
    l = (l * mask).sum(dim=1).view(2, n // 2).sum(dim=0)
    return l.view(2, n // 4).min(dim=0).values.sum() / (mask.sum() / 2)


def labelled_loss(pred, target, mask):
    """DDPMMulTrainer.backward_G, labelled branch (:225-229)."""
    l0 = ((pred[:, 0, :4] - target[:, 0, :4]) ** 2).mean(-1)
    l1 = ((pred[:, 1:] - target[:, 1:]) ** 2).mean(-1)
    l = torch.cat([l0[:, None], l1], dim=1)
    return (l * mask).sum() / mask.sum()
