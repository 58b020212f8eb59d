"""Heading / frame / finite-difference helpers used by HumanoidEnv and the reward
(mirror of the functions of uhc/utils/math_utils.py that this path calls; SURVEY.md 3.5 lists the
conventions that change numerics: heading keeps (w, z) of a wxyz quaternion, multi_quat_norm has no
abs, transform_vec_batch returns a (3, N) array)."""
import math

import numpy as np

from .transformation import (quaternion_about_axis, quaternion_inverse, quaternion_matrix, quaternion_multiply,  # noqa: F401
                             quat_mul_vec, rotation_from_quaternion)


def ewma(x, alpha=0.05):
    avg = x[0]
    for i in x[1:]:
        avg = alpha * i + (1 - alpha) * avg
    return avg


def _frame(q, trans):
    if trans == "root":
        return quaternion_matrix(q)[:3, :3]
    if trans == "heading":
        return quaternion_matrix(get_heading_q(q))[:3, :3]
    raise AssertionError(trans)


def transform_vec(v, q, trans="root"):
    """World vector -> root (or heading) frame: R(q)^T v (math_utils.py:103-115)."""
    return _frame(q, trans).T.dot(np.asarray(v, dtype=np.float64))


def transform_vec_batch(v_b, q, trans="root"):
    """(N,3) world vectors -> frame; returns shape (3, N) like the reference (math_utils.py:118-131)."""
    return _frame(q, trans).T.dot(np.asarray(v_b, dtype=np.float64).T)


def get_heading_q(q):
    hq = np.array([q[0], 0.0, 0.0, q[3]], dtype=np.float64)
    return hq / np.linalg.norm(hq)


def get_heading(q):
    hq = np.array([q[0], 0.0, 0.0, q[3]], dtype=np.float64)
    if hq[3] < 0:
        hq = -hq
    hq /= np.linalg.norm(hq)
    return 2 * math.acos(hq[0])


def de_heading(q):
    return quaternion_multiply(quaternion_inverse(get_heading_q(q)), q)


def multi_quat_diff(nq1, nq0):
    """Per-joint q1 (x) q0^-1 over flat (4N,) arrays (math_utils.py:206-216)."""
    out = np.zeros_like(nq0, dtype=np.float64)
    for i in range(nq1.shape[0] // 4):
        s = slice(4 * i, 4 * i + 4)
        out[s] = quaternion_multiply(nq1[s], quaternion_inverse(nq0[s]))
    return out


def multi_quat_norm(nq):
    return np.arccos(np.clip(nq[::4], -1.0, 1.0))


def get_angvel_fd(prev_bquat, cur_bquat, dt):
    q_diff = multi_quat_diff(cur_bquat, prev_bquat)
    n = q_diff.shape[0] // 4
    out = np.zeros(3 * n)
    for i in range(n):
        out[3 * i:3 * i + 3] = rotation_from_quaternion(q_diff[4 * i:4 * i + 4]) / dt
    return out


def get_qvel_fd_new(cur_qpos, next_qpos, dt, transform=None):
    """Finite-difference qvel between two poses (math_utils.py:45-67): root linear velocity in the world, root angular velocity = axis-angle
    of q_next (x) q_cur^-1 wrapped to (-pi, pi], divided by dt and rotated into the ROOT frame, hinge rates as angle differences wrapped the
    same way; `transform` re-expresses the linear part in the root / heading frame."""
    cur_qpos, next_qpos = np.asarray(cur_qpos, dtype=np.float64), np.asarray(next_qpos, dtype=np.float64)
    v = (next_qpos[:3] - cur_qpos[:3]) / dt
    axis, angle = rotation_from_quaternion(quaternion_multiply(next_qpos[3:7], quaternion_inverse(cur_qpos[3:7])), True)
    angle = angle - 2 * np.pi * math.ceil((angle - np.pi) / (2 * np.pi))  # into (-pi, pi]
    rv = transform_vec(axis * angle / dt, cur_qpos[3:7], "root")
    diff = next_qpos[7:] - cur_qpos[7:]
    diff = diff - 2 * np.pi * np.ceil((diff - np.pi) / (2 * np.pi))
    if transform is not None:
        v = transform_vec(v, cur_qpos[3:7], transform)
    return np.concatenate((v, rv, diff / dt))
