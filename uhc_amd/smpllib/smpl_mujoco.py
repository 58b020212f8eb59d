"""SMPL <-> simulation-model conversion tables (mirror of uhc/smpllib/smpl_mujoco.py).

``SMPLConverter`` keeps the reference's names and argument meaning
(reference: uhc/smpllib/smpl_mujoco.py:36-281) but takes this build's compiled
``Model`` objects (uhc_amd/model/mjcf.py) in place of mujoco-py models.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

_LOWER = ["L_Hip", "L_Knee", "L_Ankle", "L_Toe", "R_Hip", "R_Knee", "R_Ankle", "R_Toe"]

# per-body [kp, kd, a_scale, torque_limit]  (smpl_mujoco.py:66-91; the L/R asymmetry of
# Thorax/Shoulder torque limits is in the reference and is reproduced as is)
_BODY_PARAMS_SMPL = {
    "L_Hip": [500, 50, 1, 500], "L_Knee": [500, 50, 1, 500], "L_Ankle": [400, 40, 1, 500], "L_Toe": [200, 20, 1, 500],
    "R_Hip": [500, 50, 1, 500], "R_Knee": [500, 50, 1, 500], "R_Ankle": [400, 40, 1, 500], "R_Toe": [200, 20, 1, 500],
    "Torso": [1000, 100, 1, 500], "Spine": [1000, 100, 1, 500], "Chest": [1000, 100, 1, 500],
    "Neck": [100, 10, 1, 250], "Head": [100, 10, 1, 250],
    "L_Thorax": [400, 40, 1, 500], "L_Shoulder": [400, 40, 1, 500], "L_Elbow": [300, 30, 1, 150],
    "L_Wrist": [100, 10, 1, 150], "L_Hand": [100, 10, 1, 150],
    "R_Thorax": [400, 40, 1, 150], "R_Shoulder": [400, 40, 1, 250], "R_Elbow": [300, 30, 1, 150],
    "R_Wrist": [100, 10, 1, 150], "R_Hand": [100, 10, 1, 150],
}
# body-position difference weights (smpl_mujoco.py:40-65): toes and hands do not count
_BODY_WS_SMPL = {n: 1.0 for n in ["Pelvis"] + list(_BODY_PARAMS_SMPL)}
for _n in ("L_Toe", "R_Toe", "L_Hand", "R_Hand"):
    _BODY_WS_SMPL[_n] = 0.0


def get_body_qposaddr(model) -> "OrderedDict[str, tuple]":
    """body name -> (start, end) qpos slice (uhc/khrylib/utils/mujoco.py get_body_qposaddr)."""
    out = OrderedDict()
    for b, name in enumerate(model.body_names):
        ja, jn = model.body_jntadr[b], model.body_jntnum[b]
        if jn == 0 or ja < 0:
            continue
        start = int(model.jnt_qposadr[ja])
        end = int(model.jnt_qposadr[ja + jn]) if ja + jn < model.njnt else int(model.nq)
        out[name] = (start, end)
    return out


def get_body_qveladdr(model) -> "OrderedDict[str, tuple]":
    out = OrderedDict()
    for b, name in enumerate(model.body_names):
        ja, jn = model.body_jntadr[b], model.body_jntnum[b]
        if jn == 0 or ja < 0:
            continue
        start = int(model.jnt_dofadr[ja])
        end = int(model.jnt_dofadr[ja + jn]) if ja + jn < model.njnt else int(model.nv)
        out[name] = (start, end)
    return out


class SMPLConverter:
    def __init__(self, model, new_model, smpl_model: str = "smpl"):
        if smpl_model != "smpl":
            raise NotImplementedError("only the 24-body SMPL layout is built (SURVEY.md 8f-4)")
        self.body_ws = dict(_BODY_WS_SMPL)
        self.body_params = {k: list(v) for k, v in _BODY_PARAMS_SMPL.items()}
        self.model, self.new_model = model, new_model
        self.smpl_qpos_addr = get_body_qposaddr(model)
        self.smpl_qvel_addr = get_body_qveladdr(model)
        self.new_qpos_addr = get_body_qposaddr(new_model)
        self.new_qvel_addr = get_body_qveladdr(new_model)
        self.smpl_joint_names = list(self.smpl_qpos_addr.keys())
        self.new_joint_names = list(self.new_qpos_addr.keys())
        self.smpl_nq, self.new_nq = model.nq, new_model.nq

    def get_new_qpos_lim(self):
        a = self.new_model.jnt_qposadr
        return int(np.max(a) + a[-1] - a[-2])

    def get_new_qvel_lim(self):
        a = self.new_model.jnt_dofadr
        return int(np.max(a) + a[-1] - a[-2])

    def get_new_body_lim(self):
        return len(self.new_model.body_names)

    def get_new_diff_weight(self):
        return np.array([self.body_ws[n] if n in self.body_ws else 0 for n in self.new_joint_names])

    def _per_joint(self, col, default):
        return np.concatenate([[self.body_params[n][col]] * 3 if n in self.body_ws else [default] * 3
                               for n in self.new_joint_names[1:]]).astype(np.float64)

    def get_new_jkp(self):
        return self._per_joint(0, 50)

    def get_new_jkd(self):
        return self._per_joint(1, 5)

    def get_new_a_scale(self):
        return self._per_joint(2, 1)

    def get_new_torque_limit(self):
        return self._per_joint(3, 200)

    def qpos_new_2_smpl(self, qpos):
        subset = np.concatenate([np.arange(*self.new_qpos_addr[jt]) for jt in self.smpl_joint_names])
        return qpos[:, subset] if qpos.ndim == 2 else qpos[subset]

    def qvel_new_2_smpl(self, qvel):
        subset = np.concatenate([np.arange(*self.new_qvel_addr[jt]) for jt in self.smpl_joint_names])
        return qvel[:, subset] if qvel.ndim == 2 else qvel[subset]


# --------------------------------------------------------------------------- AMASS pose -> qpos
# SMPL joint order vs the depth-first body order of the simulation model (uhc/smpllib/smpl_parser.py:11-40)
SMPL_BONE_ORDER_NAMES = ["Pelvis", "L_Hip", "R_Hip", "Torso", "L_Knee", "R_Knee", "Spine", "L_Ankle", "R_Ankle", "Chest",
                         "L_Toe", "R_Toe", "Neck", "L_Thorax", "R_Thorax", "Head", "L_Shoulder", "R_Shoulder", "L_Elbow",
                         "R_Elbow", "L_Wrist", "R_Wrist", "L_Hand", "R_Hand"]
SMPL_BONE_KINTREE_NAMES = ["Pelvis", "L_Hip", "L_Knee", "L_Ankle", "L_Toe", "R_Hip", "R_Knee", "R_Ankle", "R_Toe", "Torso",
                           "Spine", "Chest", "Neck", "Head", "L_Thorax", "L_Shoulder", "L_Elbow", "L_Wrist", "L_Hand",
                           "R_Thorax", "R_Shoulder", "R_Elbow", "R_Wrist", "R_Hand"]
SMPL_EE_NAMES = ["L_Ankle", "R_Ankle", "L_Wrist", "R_Wrist", "Head"]
# SMPL-H: the 22 body joints of SMPL (no L_Hand / R_Hand) + 15 finger joints per hand (uhc/smpllib/smpl_parser.py:42-95)
SMPLH_BONE_ORDER_NAMES = SMPL_BONE_ORDER_NAMES[:22] + [f"{side}_{finger}{k}" for side in ("L", "R") for finger in ("Index", "Middle", "Pinky", "Ring", "Thumb")
                                                       for k in (1, 2, 3)]


def rotation_matrix_to_quaternion(R):
    """(N,3,3) -> (N,4) wxyz with the branch rule of the reference's converter
    (uhc/utils/torch_geometry_transforms.py:251-328): the branch fixes the sign of the result."""
    R = np.asarray(R, dtype=np.float64)
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    d2 = m22 < 1e-6
    d0_d1 = m00 > m11
    d0_nd1 = m00 < -m11
    q = np.zeros((R.shape[0], 4))
    c0, c1, c2, c3 = d2 & d0_d1, d2 & ~d0_d1, ~d2 & d0_nd1, ~d2 & ~d0_nd1
    t0 = 1 + m00 - m11 - m22
    t1 = 1 - m00 + m11 - m22
    t2 = 1 - m00 - m11 + m22
    t3 = 1 + m00 + m11 + m22
    cand = [
        (c0, t0, [R[:, 2, 1] - R[:, 1, 2], t0, R[:, 1, 0] + R[:, 0, 1], R[:, 0, 2] + R[:, 2, 0]]),
        (c1, t1, [R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] + R[:, 0, 1], t1, R[:, 2, 1] + R[:, 1, 2]]),
        (c2, t2, [R[:, 1, 0] - R[:, 0, 1], R[:, 0, 2] + R[:, 2, 0], R[:, 2, 1] + R[:, 1, 2], t2]),
        (c3, t3, [t3, R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]]),
    ]
    for mask, t, comps in cand:
        if mask.any():
            qq = np.stack(comps, axis=-1)[mask] / np.sqrt(t[mask])[:, None]
            q[mask] = 0.5 * qq
    return q


def smpl_to_qpose(pose, mj_model, trans=None, normalize=False, random_root=False, count_offset=True, use_quat=False,
                  euler_order="ZYX", model="smpl"):
    """AMASS axis-angle pose (B,72) [+ trans (B,3)] -> qpos (B,76) of the hinge humanoid
    (reference: uhc/smpllib/smpl_mujoco.py:543-607).  Per joint: axis-angle -> rotation -> intrinsic
    ZYX Euler angles, reordered from SMPL joint order to the model's depth-first body order; the root
    keeps a quaternion; the root position gets the model's root offset when count_offset.
    use_quat (robot.ball): every joint as a (w, x, y, z) quaternion instead -> qpos (B, 99) of the ball-joint humanoid (:590-600)."""
    from scipy.spatial.transform import Rotation as sRot
    if normalize or random_root or model != "smpl":
        raise NotImplementedError("only the options the copycat path uses are built (SURVEY.md 8f-4)")
    pose = np.asarray(pose, dtype=np.float64).reshape(-1, 72)
    B = pose.shape[0]
    if trans is None:
        trans = np.zeros((B, 3))
        trans[:, 2] = 0.91437225
    trans = np.asarray(trans, dtype=np.float64).reshape(B, 3)
    names = SMPL_BONE_ORDER_NAMES
    smpl_2_mujoco = [names.index(q) for q in get_body_qposaddr(mj_model).keys() if q in names]
    rot = sRot.from_rotvec(pose.reshape(-1, 3))
    if use_quat:
        quat = rot.as_quat()[:, [3, 0, 1, 2]].reshape(B, 24, 4)[:, smpl_2_mujoco, :].reshape(B, 96)
        qpos = np.concatenate((trans, quat), axis=1)
        if count_offset:
            qpos[:, :3] = trans + np.asarray(mj_model.body_pos)[1]
        return qpos
    eul = rot.as_euler(euler_order, degrees=False).reshape(B, 24, 3)[:, smpl_2_mujoco, :].reshape(B, 72)
    root_quat = rotation_matrix_to_quaternion(rot.as_matrix().reshape(B, 24, 3, 3)[:, 0])
    qpos = np.concatenate((trans, root_quat, eul[:, 3:]), axis=1)
    if count_offset:
        qpos[:, :3] = trans + np.asarray(mj_model.body_pos)[1]
    return qpos


def qpos_to_smpl(qpos, mj_model, smpl_model="smpl"):
    """Inverse of `smpl_to_qpose` for hinge models (smpl_mujoco.py:738-752): qpos (T, nq) -> (pose_aa (T, J, 3) in SMPL joint order, trans (T, 3)).
    Root: the wxyz quaternion as a rotation vector; every other joint: its (z, y, x) hinge triple read as intrinsic ZYX Euler angles."""
    from scipy.spatial.transform import Rotation as sRot
    qpos = np.asarray(qpos, dtype=np.float64)
    addr = get_body_qposaddr(mj_model)
    names = SMPL_BONE_ORDER_NAMES if smpl_model == "smpl" else SMPLH_BONE_ORDER_NAMES
    pose = np.zeros((qpos.shape[0], len(names), 3))
    pose[:, 0] = sRot.from_quat(qpos[:, [4, 5, 6, 3]]).as_rotvec()
    for k, name in enumerate(names[1:], start=1):
        a, b = addr[name]
        pose[:, k] = sRot.from_euler("ZYX", qpos[:, a:b]).as_rotvec()
    return pose, qpos[:, :3] - np.asarray(mj_model.body_pos)[1]


def smpl_6d_to_qpose(full_pose, model, normalize=False):
    """[trans (3) | 24 x 6D rotations] -> qpos (smpl_mujoco.py:776-780): the 6D blocks are the first two COLUMNS of each rotation matrix
    (process_amass_db.convert_aa_to_orth6d); Gram-Schmidt gives the matrix back, its rotation vector goes through `smpl_to_qpose`."""
    from scipy.spatial.transform import Rotation as sRot
    full_pose = np.asarray(full_pose, dtype=np.float64)
    d6 = full_pose[:, 3:].reshape(full_pose.shape[0], -1, 6)  # per joint: column 0 of R, then column 1
    a, b = d6[..., 0:3], d6[..., 3:6]
    x = a / np.linalg.norm(a, axis=-1, keepdims=True)
    z = np.cross(x, b)
    z /= np.linalg.norm(z, axis=-1, keepdims=True)
    R = np.stack([x, np.cross(z, x), z], axis=-1)
    pose_aa = sRot.from_matrix(R.reshape(-1, 3, 3)).as_rotvec().reshape(full_pose.shape[0], -1)
    return smpl_to_qpose(pose_aa, model, trans=full_pose[:, :3], normalize=normalize)
