"""ORACLE helper (test infrastructure): deterministic parameter / input generators.

Every tensor is a pure function of its NAME and SHAPE (numpy legacy MT19937 seeded by
crc32(name) -- bit-stable across numpy versions), so goldens made from the reference in
the build container can be re-created on the GPU box without shipping 81 M parameters.
The zero-initialised tensors of the reference (transformer.py:51-57,72,162,378) are
overwritten too; otherwise every parity check is 0 == 0.
"""
import zlib

import numpy as np
import torch


def _rs(name):
    return np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)


_CACHE, _CACHE_BYTES, _CACHE_CAP = {}, [0], 3 << 30


def tensor_for(name, shape, dtype=torch.float32):
    """Pure function of (name, shape): the large tensors are memoised (the test-suite rebuilds the same 81 M-parameter model
    dozens of times; 3 GiB cap, oldest entries dropped) and handed out as copies."""
    shape = tuple(shape)
    key = (name, shape, dtype)
    if key in _CACHE:
        return _CACHE[key].clone()
    t = _tensor_for(name, shape, dtype)
    if t.numel() >= 1 << 16:
        _CACHE[key] = t
        _CACHE_BYTES[0] += t.numel() * t.element_size()
        while _CACHE_BYTES[0] > _CACHE_CAP:
            old = _CACHE.pop(next(iter(_CACHE)))
            _CACHE_BYTES[0] -= old.numel() * old.element_size()
        return t.clone()
    return t


def _tensor_for(name, shape, dtype):
    n = _rs(name).standard_normal(shape)
    if name == "sequence_embedding":
        v = n
    elif name.endswith("norm.weight") or name.endswith("ln.weight"):
        v = 1.0 + 0.1 * n
    elif name.endswith(".bias"):
        v = 0.1 * n
    elif len(shape) == 2:
        v = n / np.sqrt(shape[1])  # (out, in): unit-variance outputs
    else:
        v = 0.1 * n
    return torch.from_numpy(np.ascontiguousarray(v)).to(dtype)


def fill_state_dict(sd, dtype=torch.float32, skip_prefixes=("clip.",)):
    """Returns a new {name: tensor} with every entry replaced by tensor_for(name)."""
    out = {}
    for k, v in sd.items():
        if any(k.startswith(s) for s in skip_prefixes):
            out[k] = v.detach().clone()
        else:
            out[k] = tensor_for(k, v.shape, dtype)
    return out


def core_param_shapes(F, d, ff, L, Lt, num_frames):
    """Names/shapes of the denoiser core, without CLIP / text head."""
    E = 4 * d
    s = {
        "sequence_embedding": (num_frames, d),
        "joint_embed.weight": (d, F), "joint_embed.bias": (d,),
        "time_embed.0.weight": (E, d), "time_embed.0.bias": (E,),
        "time_embed.2.weight": (E, E), "time_embed.2.bias": (E,),
        "out.weight": (F, d), "out.bias": (F,),
    }

    def sty(pre):
        s[pre + ".emb_layers.1.weight"] = (2 * d, E)
        s[pre + ".emb_layers.1.bias"] = (2 * d,)
        s[pre + ".norm.weight"] = (d,)
        s[pre + ".norm.bias"] = (d,)
        s[pre + ".out_layers.2.weight"] = (d, d)
        s[pre + ".out_layers.2.bias"] = (d,)

    for l in range(L):
        b = "temporal_decoder_blocks.%d" % l
        for blk in ("sa_block", "ca_block"):
            s[b + "." + blk + ".norm.weight"] = (d,)
            s[b + "." + blk + ".norm.bias"] = (d,)
            kin = d if blk == "sa_block" else Lt
            s[b + "." + blk + ".query.weight"] = (d, d)
            s[b + "." + blk + ".query.bias"] = (d,)
            for nm in ("key", "value"):
                s[b + "." + blk + "." + nm + ".weight"] = (d, kin)
                s[b + "." + blk + "." + nm + ".bias"] = (d,)
            sty(b + "." + blk + ".proj_out")
        s[b + ".ca_block.text_norm.weight"] = (Lt,)
        s[b + ".ca_block.text_norm.bias"] = (Lt,)
        s[b + ".ffn.linear1.weight"] = (ff, d)
        s[b + ".ffn.linear1.bias"] = (ff,)
        s[b + ".ffn.linear2.weight"] = (d, ff)
        s[b + ".ffn.linear2.bias"] = (d,)
        sty(b + ".ffn.proj_out")
    return s


def interaction_param_shapes(F, d, ff, L, Lt, num_frames, no_cross_attn=False):
    """Denoiser core of MotionInteractionTransformer (interaction_transformer.py:459-507): the
    single-person core + joint_embed2, out2 and one int_ca_block per layer."""
    s = core_param_shapes(F, d, ff, L, Lt, num_frames)
    E = 4 * d
    s["joint_embed2.weight"] = (d, 4)
    s["joint_embed2.bias"] = (d,)
    s["out2.weight"] = (F, d)
    s["out2.bias"] = (F,)
    if not no_cross_attn:
        for l in range(L):
            b = "temporal_decoder_blocks.%d.int_ca_block" % l
            s[b + ".norm.weight"] = (d,)
            s[b + ".norm.bias"] = (d,)
            for nm in ("query", "key", "value"):
                s[b + "." + nm + ".weight"] = (d, d)
                s[b + "." + nm + ".bias"] = (d,)
            s[b + ".proj_out.emb_layers.1.weight"] = (2 * d, E)
            s[b + ".proj_out.emb_layers.1.bias"] = (2 * d,)
            s[b + ".proj_out.norm.weight"] = (d,)
            s[b + ".proj_out.norm.bias"] = (d,)
            s[b + ".proj_out.out_layers.2.weight"] = (d, d)
            s[b + ".proj_out.out_layers.2.bias"] = (d,)
    return s


def interaction_params(F, d, ff, L, Lt, num_frames, dtype=torch.float32, no_cross_attn=False):
    return {k: tensor_for(k, v, dtype)
            for k, v in interaction_param_shapes(F, d, ff, L, Lt, num_frames, no_cross_attn).items()}


def core_params(F, d, ff, L, Lt, num_frames, dtype=torch.float32):
    return {k: tensor_for(k, v, dtype) for k, v in core_param_shapes(F, d, ff, L, Lt, num_frames).items()}


def inputs(B, T, F, d, N, Lt, lengths, t_values, dtype=torch.float32):
    """Seeded-by-name inputs: x_t, xf_proj, xf_out ~ N(0,1); t and length given."""
    tag = "B%dT%dF%dd%d" % (B, T, F, d)
    return {
        "x": tensor_for("in.x." + tag, (B, T, F), dtype) * 10.0,          # 0.1*n*10 = N(0,1)
        "xf_proj": tensor_for("in.xf_proj." + tag, (B, 4 * d), dtype) * 10.0,
        "xf_out": tensor_for("in.xf_out." + tag, (B, N, Lt), dtype) * 10.0,
        "t": torch.tensor(list(t_values), dtype=torch.int64),
        "length": torch.tensor(list(lengths), dtype=torch.int64),
    }


# two-person cases: B = number of PAIRS (the model batch is 2B); lengths/t are per pair
ICASES = {
    "tiny2": dict(B=2, T=16, F=12, d=64, H=8, L=2, ff=128, N=77, Lt=32, num_frames=20,
                  lengths=(16, 9), t=(3, 987)),
    "config1x2": dict(B=2, T=61, F=150, d=128, H=8, L=3, ff=1024, N=77, Lt=256, num_frames=60,
                      lengths=(61, 41), t=(0, 500)),
}

CASES = {
    # name: dict(B,T,F,d,H,L,ff,N,Lt,num_frames,lengths,t)
    "tiny": dict(B=2, T=16, F=12, d=64, H=8, L=2, ff=128, N=77, Lt=32, num_frames=20,
                 lengths=(16, 9), t=(3, 987)),
    "config1": dict(B=2, T=60, F=150, d=128, H=8, L=4, ff=1024, N=77, Lt=256, num_frames=60,
                    lengths=(60, 41), t=(0, 500)),
    "width": dict(B=2, T=196, F=150, d=512, H=8, L=8, ff=1024, N=77, Lt=256, num_frames=196,
                  lengths=(196, 77), t=(999, 250)),
}
