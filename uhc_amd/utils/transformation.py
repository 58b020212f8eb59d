"""Quaternion helpers (w, x, y, z) used by the env / reward / data path.

Same names and conventions as the functions the reference takes from uhc/utils/transformation.py
(quaternion_* : lines 1232-1534, rotation_from_quaternion : 362-373) so that callers read alike;
the implementations are this build's own (vectorised where the reference loops)."""
import math

import numpy as np

_EPS = np.finfo(float).eps * 4.0


def quaternion_multiply(q1, q0):
    """q1 (x) q0 (transformation.py:1478-1492)."""
    w1, x1, y1, z1 = q1
    w0, x0, y0, z0 = q0
    return np.array([w1 * w0 - x1 * x0 - y1 * y0 - z1 * z0,
                     w1 * x0 + x1 * w0 + y1 * z0 - z1 * y0,
                     w1 * y0 - x1 * z0 + y1 * w0 + z1 * x0,
                     w1 * z0 + x1 * y0 - y1 * x0 + z1 * w0], dtype=np.float64)


def quaternion_multiply_batch(q0, q1):
    """Row-wise q0 (x) q1 (transformation.py:1455-1475)."""
    q0, q1 = np.asarray(q0, dtype=np.float64), np.asarray(q1, dtype=np.float64)
    w0, x0, y0, z0 = q0[..., 0], q0[..., 1], q0[..., 2], q0[..., 3]
    w1, x1, y1, z1 = q1[..., 0], q1[..., 1], q1[..., 2], q1[..., 3]
    return np.stack([w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1,
                     w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
                     w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1,
                     w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1], axis=-1)


def quaternion_inverse(q):
    """conj(q) / |q|^2 (transformation.py:1509-1520) -- NOT assuming a unit quaternion."""
    q = np.array(q, dtype=np.float64)
    return np.array([q[0], -q[1], -q[2], -q[3]]) / np.dot(q, q)


def quaternion_inverse_batch(q):
    q = np.asarray(q, dtype=np.float64)
    c = q * np.array([1.0, -1.0, -1.0, -1.0])
    return c / np.linalg.norm(q, axis=-1, keepdims=True)  # by the NORM, not its square (transformation.py:1532-1534): differs from quaternion_inverse off the unit sphere


def quaternion_matrix(q):
    """4x4 homogeneous rotation of a (not necessarily unit) quaternion (transformation.py:1344-1368)."""
    q = np.array(q, dtype=np.float64)
    n = np.dot(q, q)
    M = np.identity(4)
    if n < _EPS:
        return M
    w, x, y, z = q * math.sqrt(2.0 / n)
    M[:3, :3] = [[1.0 - y * y - z * z, x * y - z * w, x * z + y * w],
                 [x * y + z * w, 1.0 - x * x - z * z, y * z - x * w],
                 [x * z - y * w, y * z + x * w, 1.0 - x * x - y * y]]
    return M


def quaternion_about_axis(angle, axis):
    """transformation.py:347-359: the axis is used as given (not normalised)."""
    s = math.sin(angle / 2.0)
    return np.array([math.cos(angle / 2.0), axis[0] * s, axis[1] * s, axis[2] * s], dtype=np.float64)


def quaternion_from_euler_rzyx(az, ay, ax):
    """quaternion_from_euler(az, ay, ax, 'rzyx') (transformation.py:1232-1285): intrinsic rotations about
    z, then the new y, then the new x, i.e. qz (x) qy (x) qx.  Accepts arrays."""
    az, ay, ax = np.asarray(az, dtype=np.float64) / 2, np.asarray(ay, dtype=np.float64) / 2, np.asarray(ax, dtype=np.float64) / 2
    cz, sz, cy, sy, cx, sx = np.cos(az), np.sin(az), np.cos(ay), np.sin(ay), np.cos(ax), np.sin(ax)
    return np.stack([cx * cy * cz + sx * sy * sz,
                     sx * cy * cz - cx * sy * sz,
                     cx * sy * cz + sx * cy * sz,
                     cx * cy * sz - sx * sy * cz], axis=-1)


def quaternion_from_euler(ai, aj, ak, axes="rzyx"):
    if axes != "rzyx":
        raise NotImplementedError("only the 'rzyx' convention is used on this path (humanoid_im.py:943)")
    return quaternion_from_euler_rzyx(ai, aj, ak)


def rotation_from_quaternion(q, separate=False):
    """axis * angle of a unit quaternion; zero when |1 -/+ w| < 1e-6, angle in [0, 2pi) (transformation.py:362-373)."""
    if abs(1.0 - q[0]) < 1e-6 or abs(1.0 + q[0]) < 1e-6:
        axis, angle = np.array([1.0, 0.0, 0.0]), 0.0
    else:
        angle = 2 * math.acos(q[0])
        axis = np.asarray(q[1:4], dtype=np.float64) / math.sin(angle / 2.0)
        axis = axis / np.linalg.norm(axis)
    return (axis, angle) if separate else axis * angle


def quat_mul_vec(q, v):
    return quaternion_matrix(q)[:3, :3].dot(np.asarray(v, dtype=np.float64))


def quaternion_from_euler_batch(ai, aj, ak, axes="rzyx"):
    """(B,) or (B, 1) angle arrays -> (B, 4) wxyz quaternions (transformation.py:1288-1345), for the one convention this path uses."""
    if axes != "rzyx":
        raise NotImplementedError("only the 'rzyx' convention is used on this path (humanoid_im.py:943)")
    return quaternion_from_euler_rzyx(np.asarray(ai).reshape(-1), np.asarray(aj).reshape(-1), np.asarray(ak).reshape(-1))


def quat_mul_vec_batch(q, v):
    """Rotate v (*, 3) by the unit quaternions q (*, 4): v + 2 (w u x v + u x (u x v)), u = the vector part (transformation.py:1212-1229)."""
    q, v = np.asarray(q, dtype=np.float64), np.asarray(v, dtype=np.float64)
    assert q.shape[-1] == 4 and v.shape[-1] == 3 and q.shape[:-1] == v.shape[:-1]
    u = q[..., 1:]
    uv = np.cross(u, v)
    return v + 2.0 * (q[..., :1] * uv + np.cross(u, uv))
