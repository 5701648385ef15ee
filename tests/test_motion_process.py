"""Post-sampling joint recovery: oracle vs the reference's own outputs (golden g12) on
the CPU; HIP kernel vs golden / oracle on the GPU.  Floating point: 1e-5 of the position scale."""
import numpy as np
import pytest
import torch

import hig_amd  # noqa: F401
from hig_amd.utils import motion_process as MP
from oracle import fill
from oracle import motion_ref as MR

B, T, F, J = 3, 90, 263, 22


def inputs():
    d1 = fill.tensor_for("g12.data1", (B, T + 1, F)) * 10.0
    d2 = fill.tensor_for("g12.data2", (B, T + 1, F)) * 10.0
    for d in (d1, d2):
        d[..., 0] *= 0.05
        d[:, -1, 2:4] = torch.nn.functional.normalize(d[:, -1, 2:4], dim=-1)
    return d1, d2


def test_oracle_matches_reference_golden(gold):
    g = gold("g12_recover.npz")
    d1, d2 = inputs()
    p1, p2 = MR.recover_from_ric2(d1, d2, J)
    assert np.array_equal(p1.numpy(), g["pos1"]) and np.array_equal(p2.numpy(), g["pos2"])
    q, r = MR.recover_root_rot_pos(d1[:, :-1])
    assert np.array_equal(q.numpy(), g["quat"]) and np.array_equal(r.numpy(), g["rpos"])


def test_host_tensors_are_refused():
    d1, d2 = inputs()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        MP.recover_from_ric2(d1, d2, J)


@pytest.mark.gpu
def test_hip_recover_matches_reference_golden(gold):
    g = gold("g12_recover.npz")
    d1, d2 = inputs()
    p1, p2 = MP.recover_from_ric2(d1.cuda(), d2.cuda(), J)
    scale = np.abs(g["pos1"]).max()
    assert p1.shape == (B, T, J, 3)
    assert np.abs(p1.cpu().numpy() - g["pos1"]).max() < 1e-5 * scale
    assert np.abs(p2.cpu().numpy() - g["pos2"]).max() < 1e-5 * scale


@pytest.mark.gpu
def test_fused_denormalise_and_recover_from_sampler_layout():
    """Sampler layout (token 0 = init row, Z-normalised) through the fused path == oracle's
    denormalise -> reorder -> recover; also a long clip (T = 300) and a 21-joint skeleton."""
    for (T2, J2, R) in ((90, 22, 8), (300, 21, 3)):
        x = fill.tensor_for("mp.x.%d" % T2, (R, T2 + 1, F)) * 10.0
        x[..., 0] *= 0.05
        mean = fill.tensor_for("mp.mean", (F,)) * 10.0
        std = 1.0 + 0.5 * (fill.tensor_for("mp.std", (F,)) * 10.0).abs()
        imean = fill.tensor_for("mp.imean", (4,)) * 10.0
        istd = 1.0 + 0.5 * (fill.tensor_for("mp.istd", (4,)) * 10.0).abs()
        got = MP.generated_to_joints(x.cuda(), mean.numpy(), std.numpy(), imean.numpy(), istd.numpy(), J2).cpu()
        den = torch.stack([MR.denormalize_sample(x[r].double(), mean.double(), std.double(), imean.double(),
                                                 istd.double()) for r in range(R)])
        ref = MR.recover_person(den, J2)
        assert got.shape == (R, T2, J2, 3)
        assert (got.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
