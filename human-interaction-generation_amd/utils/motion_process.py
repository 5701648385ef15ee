"""Post-sampling consumers on the device: the reference's `recover_from_ric2`
(codes/utils/motion_process.py:418-462) and the de-normalisation in front of it
(codes/tools/visualization.py:146-152) as one HIP kernel (`hig_recover_joints`), so a generated batch
goes from the sampler's (2B, T + 1, F) tokens to world-space joints without leaving HBM.
"""
import numpy as np
import torch

from .. import _lib


def _launch(motion, stats, joints_num, init_first):
    if not motion.is_cuda:
        raise RuntimeError("recover_from_ric2: ROCm device tensors required (no CPU fallback)")
    motion = motion.float().contiguous()
    R, T1, F = motion.shape
    pos = torch.empty(R, T1 - 1, joints_num, 3, device=motion.device, dtype=torch.float32)
    _lib.check(_lib.lib().hig_recover_joints(_lib.ptr(motion), _lib.ptr(stats), R, T1 - 1, F, joints_num,
                                             int(init_first), _lib.ptr(pos), _lib.stream_ptr()))
    return pos


def recover_from_ric2(data1, data2, joints_num):
    """Reference signature: de-normalised motions (B, T + 1, F) with the init state in the LAST row
    -> (positions1, positions2), each (B, T, joints_num, 3).  (The reference itself only broadcasts
    correctly for B == 1; every sample here is treated the way its B == 1 call treats one.)"""
    B = data1.shape[0]
    pos = _launch(torch.cat([data1, data2], dim=0), None, joints_num, init_first=False)
    return pos[:B], pos[B:]


def generated_to_joints(samples, mean, std, init_mean, init_std, joints_num=22):
    """Sampler output (R, T + 1, F), token 0 = init-pose row, Z-normalised -> joints (R, T, J, 3):
    visualization.py:146-152 + recover_from_ric2 fused."""
    dev = samples.device
    stats = torch.cat([torch.as_tensor(np.asarray(a), dtype=torch.float32).reshape(-1)
                       for a in (mean, std, init_mean, init_std)]).to(dev)
    assert stats.numel() == 2 * samples.shape[2] + 8, "mean/std must have F entries, init_mean/init_std 4"
    return _launch(samples, stats, joints_num, init_first=True)
