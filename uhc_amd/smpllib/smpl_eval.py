"""Imitation metrics (mirror of uhc/smpllib/smpl_eval.py:24-126): mpjpe, global mpjpe, Procrustes-aligned
mpjpe, velocity / acceleration error (all in mm), root transform distance, success.  Vectorised numpy."""
import numpy as np

from ..utils.transformation import quaternion_matrix


def p_mpjpe(predicted, target):
    """MPJPE after similarity (scale, rotation, translation) alignment per frame (smpl_eval.py:24-63)."""
    muX, muY = target.mean(axis=1, keepdims=True), predicted.mean(axis=1, keepdims=True)
    X0, Y0 = target - muX, predicted - muY
    normX = np.sqrt((X0 ** 2).sum(axis=(1, 2), keepdims=True))
    normY = np.sqrt((Y0 ** 2).sum(axis=(1, 2), keepdims=True))
    X0, Y0 = X0 / normX, Y0 / normY
    U, s, Vt = np.linalg.svd(np.matmul(X0.transpose(0, 2, 1), Y0))
    V = Vt.transpose(0, 2, 1)
    R = np.matmul(V, U.transpose(0, 2, 1))
    sign = np.sign(np.linalg.det(R))[:, None]
    V[:, :, -1] *= sign
    s[:, -1] *= sign.ravel()
    R = np.matmul(V, U.transpose(0, 2, 1))
    a = s.sum(axis=1)[:, None, None] * normX / normY
    t = muX - a * np.matmul(muY, R)
    return np.linalg.norm(a * np.matmul(predicted, R) + t - target, axis=-1)


def _root_matrices(qpos):
    out = np.stack([quaternion_matrix(q[3:7]) for q in qpos])
    out[:, :3, 3] = qpos[:, :3]
    return out


def compute_metrics(res, converter=None):
    """res: {"pred", "gt": (T, nq) qpos; "pred_jpos", "gt_jpos": (T, 3J) world joint positions; "fail_safe", "percent"}."""
    jp, jg = np.asarray(res["pred_jpos"]), np.asarray(res["gt_jpos"])
    T = np.asarray(res["pred"]).shape[0]
    jp, jg = jp.reshape(T, -1, 3), jg.reshape(T, -1, 3)
    Mp, Mg = _root_matrices(np.asarray(res["pred"])), _root_matrices(np.asarray(res["gt"]))
    err = np.eye(4)[None] - np.matmul(Mp, np.linalg.inv(Mg))
    out = {"root_dist": np.sqrt((err ** 2).sum(axis=(1, 2))) / T * 1000}
    vel = np.linalg.norm((jg[1:] - jg[:-1]) - (jp[1:] - jp[:-1]), axis=2)  # argument order of the reference: (pred, gt) swapped is symmetric
    acc = np.linalg.norm((jg[:-2] - 2 * jg[1:-1] + jg[2:]) - (jp[:-2] - 2 * jp[1:-1] + jp[2:]), axis=2)
    out["vel_dist"] = vel.mean(axis=-1) * 1000
    out["accel_dist"] = acc.mean(axis=-1) * 1000
    out["mpjpe_g"] = np.linalg.norm(jp - jg, axis=2).mean(axis=-1) * 1000
    root = 0 if jp.shape[1] == 24 else 7
    jp0, jg0 = jp - jp[:, root:root + 1], jg - jg[:, root:root + 1]
    out["pa_mpjpe"] = p_mpjpe(jp0, jg0).mean(axis=-1) * 1000
    out["mpjpe"] = np.linalg.norm(jp0 - jg0, axis=2).mean(axis=-1) * 1000
    out["succ"] = np.array([(not res["fail_safe"]) and res["percent"] == 1])
    return out
