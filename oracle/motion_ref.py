"""ORACLE (test infrastructure, never the product path).

CPU restatement of the post-sampling joint recovery of the two-person pipeline
(reference: codes/utils/motion_process.py:362-382 `recover_root_rot_pos`, :418-462
`recover_from_ric2`, codes/utils/quaternion.py:16-20,54-73 `qinv` / `qrot`) and of the
de-normalisation that precedes it (codes/tools/visualization.py:146-152).
Pinned by tests/golden/g12_recover.npz (outputs of the reference functions themselves).
"""
import torch


def qrot(q, v):
    qvec = q[..., 1:]
    uv = torch.cross(qvec, v, dim=-1)
    uuv = torch.cross(qvec, uv, dim=-1)
    return v + 2 * (q[..., :1] * uv + uuv)


def qinv(q):
    return q * torch.tensor([1.0, -1.0, -1.0, -1.0], dtype=q.dtype)


def recover_root_rot_pos(data):
    """data (..., T, F): feature 0 = root angular velocity about Y, 1:3 = root XZ velocity (in the
    root frame), 3 = root height -> per-frame root rotation quaternion (w,0,y,0) and position."""
    rot_vel = data[..., 0]
    ang = torch.zeros_like(rot_vel)
    ang[..., 1:] = rot_vel[..., :-1]
    ang = torch.cumsum(ang, dim=-1)
    quat = torch.zeros(data.shape[:-1] + (4,), dtype=data.dtype)
    quat[..., 0] = torch.cos(ang)
    quat[..., 2] = torch.sin(ang)
    pos = torch.zeros(data.shape[:-1] + (3,), dtype=data.dtype)
    pos[..., 1:, [0, 2]] = data[..., :-1, 1:3]
    pos = qrot(qinv(quat), pos)
    pos = torch.cumsum(pos, dim=-2)
    pos[..., 1] = data[..., 3]
    return quat, pos


def recover_person(data, joints_num):
    """One person of recover_from_ric2: data (B, T + 1, F) with the init state in the LAST row
    ([x, z, quat_w, quat_y, ...]) -> joint positions (B, T, joints_num, 3)."""
    body, init = data[:, :-1], data[:, -1]
    quat, rpos = recover_root_rot_pos(body)
    p = body[..., 4:(joints_num - 1) * 3 + 4]
    p = p.reshape(p.shape[:-1] + (-1, 3))
    p = qrot(qinv(quat[..., None, :]).expand(p.shape[:-1] + (4,)), p)
    p = p + torch.stack([rpos[..., 0:1], torch.zeros_like(rpos[..., 0:1]), rpos[..., 2:3]], dim=-1)
    p = torch.cat([rpos.unsqueeze(-2), p], dim=-2)
    q0 = torch.zeros(init.shape[0], 4, dtype=data.dtype)
    q0[:, 0], q0[:, 2] = init[:, 2], init[:, 3]
    p = qrot(q0[:, None, None, :].expand(p.shape[:-1] + (4,)), p)
    p[..., 0] += init[:, None, None, 0]
    p[..., 2] += init[:, None, None, 1]
    return p


def recover_from_ric2(data1, data2, joints_num):
    return recover_person(data1, joints_num), recover_person(data2, joints_num)


def denormalize_sample(m, mean, std, init_mean, init_std):
    """visualization.py:146-152 for one generated motion (T + 1, F) whose row 0 is the init-pose token:
    undo the Z-norm and move the init row to the end (the layout recover_from_ric2 expects)."""
    m = m.clone()
    m[1:] = m[1:] * std + mean
    m[0, :4] = m[0, :4] * init_std + init_mean
    return torch.cat([m[1:], m[0][None]], dim=0)
