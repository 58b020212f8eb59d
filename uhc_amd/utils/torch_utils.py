"""Batched torch quaternion helpers for the expert-feature path (the subset of
uhc/utils/torch_utils.py that Humanoid.qpos_fk calls; own implementations, same conventions)."""
import math

import torch


def quaternion_from_euler_rzyx(az, ay, ax):
    """torch_utils.py:64-118 with axes='rzyx': qz (x) qy (x) qx."""
    az, ay, ax = az / 2, ay / 2, ax / 2
    cz, sz, cy, sy, cx, sx = torch.cos(az), torch.sin(az), torch.cos(ay), torch.sin(ay), torch.cos(ax), torch.sin(ax)
    return torch.stack([cx * cy * cz + sx * sy * sz, sx * cy * cz - cx * sy * sz,
                        cx * sy * cz + sx * cy * sz, cx * cy * sz - sx * sy * cz], dim=-1)


def quaternion_multiply_batch(q0, q1):
    """q0 (x) q1, any leading shape (torch_utils.py:408-429)."""
    w0, x0, y0, z0 = q0.unbind(-1)
    w1, x1, y1, z1 = q1.unbind(-1)
    return torch.stack([w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1, w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
                        w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1, w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1], dim=-1)


def quaternion_inverse_batch(q):
    sign = torch.tensor([1.0, -1.0, -1.0, -1.0], dtype=q.dtype, device=q.device)
    return q * sign / (q * q).sum(-1, keepdim=True)


def quat_mul_vec_batch(q, v):
    """Rotate v by q with the cross-product form (torch_utils.py:490-508); assumes |q| = 1 like the reference."""
    qv = q[..., 1:]
    uv = torch.cross(qv, v, dim=-1)
    uuv = torch.cross(qv, uv, dim=-1)
    return v + 2 * (q[..., :1] * uv + uuv)


def quaternion_matrix_batch(q):
    """(B,4) -> (B,3,3), normalising first (torch_utils.py:191-217)."""
    q = q / torch.norm(q, dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz, txx, txy, txz, tyy, tyz, tzz = tx * w, ty * w, tz * w, tx * x, ty * x, tz * x, ty * y, tz * y, tz * z
    return torch.stack([torch.stack([1 - (tyy + tzz), txy - twz, txz + twy], -1),
                        torch.stack([txy + twz, 1 - (txx + tzz), tyz - twx], -1),
                        torch.stack([txz - twy, tyz + twx, 1 - (txx + tyy)], -1)], -2)


def transform_vec_batch(v, q):
    """World vectors -> root frame: R(q)^T v (torch_utils.py:335-345, trans='root')."""
    return torch.matmul(quaternion_matrix_batch(q).transpose(-1, -2), v.unsqueeze(-1)).squeeze(-1)


def safe_acos(x):
    return torch.acos(torch.clamp(x, -1.0 + 1e-7, 1.0 - 1e-7))


def rotation_from_quaternion_batch(q, separate=False):
    """axis * angle with the zero-rotation guard |sin(acos w)| < 1e-5 (torch_utils.py:142-167)."""
    half = safe_acos(q[..., 0])
    s = torch.sin(half)
    cond = s.abs() < 1e-5
    zero_axis = torch.zeros_like(q[..., 1:])
    zero_axis[..., 0] = 1.0
    axis = torch.where(cond.unsqueeze(-1), zero_axis, q[..., 1:] / torch.where(cond, torch.ones_like(s), s).unsqueeze(-1))
    angle = torch.where(cond, torch.zeros_like(half), 2 * half)
    return (axis, angle) if separate else axis * angle.unsqueeze(-1)


def get_qvel_fd_batch(cur_qpos, next_qpos, dt):
    """Finite-difference generalised velocity (torch_utils.py:368-386): root angular part is the axis-angle of
    q_next (x) q_cur^-1 wrapped to (-pi, pi], expressed in the root frame; joints are plain differences."""
    v = (next_qpos[:, :3] - cur_qpos[:, :3]) / dt
    qrel = quaternion_multiply_batch(next_qpos[:, 3:7], quaternion_inverse_batch(cur_qpos[:, 3:7]))
    axis, angle = rotation_from_quaternion_batch(qrel, True)
    angle = torch.where(angle > math.pi, angle - 2 * math.pi, angle)
    angle = torch.where(angle < -math.pi, angle + 2 * math.pi, angle)
    rv = transform_vec_batch(axis * angle.unsqueeze(-1) / dt, cur_qpos[:, 3:7])
    return torch.cat((v, rv, (next_qpos[:, 7:] - cur_qpos[:, 7:]) / dt), dim=1)


def get_angvel_fd_batch(prev_bquat, cur_bquat, dt):
    """(T,J,4) x2 -> (T,J,3) (torch_utils.py:121-126)."""
    qd = quaternion_multiply_batch(cur_bquat, quaternion_inverse_batch(prev_bquat))
    return rotation_from_quaternion_batch(qd) / dt
