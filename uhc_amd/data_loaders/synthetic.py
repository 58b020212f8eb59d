"""Synthetic AMASS-format clips (the real AMASS pickles and SMPL files are licensed and absent):
the shipped standing pose with smooth seeded joint-space perturbations, random heading, 150-300 frames
(SURVEY.md 8d, config 2).  Output matches what DatasetAMASSSingle consumes."""
import os

import numpy as np
from scipy.spatial.transform import Rotation as sRot

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def make_synthetic_amass(n_clips=64, seed=1, t_range=(150, 300), amp=0.3, root_height=0.91437225, model_root_z=0.0282, objects=0, obj_seed=11):
    """objects: K > 0 adds `obj_pose` (T, 7 K) to every clip -- where a reset puts K free objects (the reference's GRAB clips carry the
    objects' recorded trajectory there; a reset reads the window's first row, uhc/envs/humanoid_im.py:1284-1287): K poses on a circle of
    0.75 m around the humanoid, stacked in height (objects dropped around each humanoid, default_rng(11))."""
    z = np.load(os.path.join(_ASSETS, "standing_neutral.npz"))
    base = z["pose_aa"][10].copy()
    rng = np.random.default_rng(seed)
    out = {}
    for c in range(n_clips):
        T = int(rng.integers(t_range[0], t_range[1] + 1))
        t = np.arange(T)[:, None] / 30.0
        a = rng.uniform(0, amp, size=(1, 72))
        f = rng.uniform(0.2, 2.0, size=(1, 72))
        ph = rng.uniform(0, 2 * np.pi, size=(1, 72))
        pose = base[None] + a * (np.sin(2 * np.pi * f * t + ph) - np.sin(ph))  # starts exactly at the standing pose
        yaw = rng.uniform(-np.pi, np.pi)
        root = (sRot.from_euler("z", yaw) * sRot.from_rotvec(base[:3])).as_rotvec()
        pose[:, :3] = root
        trans = np.zeros((T, 3))
        trans[:, 2] = root_height - model_root_z
        R = sRot.from_rotvec(pose.reshape(-1, 3)).as_matrix().reshape(T, 24, 3, 3)
        pose_6d = R[..., :2].transpose(0, 1, 3, 2).reshape(T, 144)
        out[f"0-synth_{c:04d}"] = {"pose_aa": pose, "pose_6d": pose_6d, "trans": trans, "beta": rng.normal(scale=0.5, size=10),
                                  "gender": "neutral", "seq_name": f"0-synth_{c:04d}"}
        if objects:
            orng = np.random.default_rng(obj_seed + c)
            ang = orng.uniform(0, 2 * np.pi, size=objects)
            row = np.concatenate([np.r_[-0.15 + 0.75 * np.cos(a), -0.05 + 0.75 * np.sin(a), 0.3 + 0.45 * k, 1.0, 0.0, 0.0, 0.0] for k, a in enumerate(ang)])
            out[f"0-synth_{c:04d}"]["obj_pose"] = np.tile(row, (T, 1))
    return out


def make_synthetic_body_provider(vary_hulls=True):
    """A stand-in for the SMPL forward pass (licensed model files: absent) with the same interface as uhc_amd.smpllib.smpl_robot.SMPLBody:
    (betas, gender) -> (vertices (V, 3), joints (24, 3), skin weights (V, 24)), everything in the SMPL rest pose and joint order.
    The "body" is the shipped neutral asset's own hulls; beta[0] scales the stature (all joint offsets), beta[1] the girth (vertices about
    their joint), gender shifts the girth a little; with `vary_hulls`, beta[2] drops a few vertices of the hands and feet, so that
    different betas give hulls of different sizes (what real shapes do after decimation).  Feeds the shape -> model generator in tests
    and in `bench.py`; it is NOT an SMPL model."""
    from ..model.mjcf import kinematics_np, quat_to_mat
    from ..sim import load_asset_model
    from ..smpllib.smpl_mujoco import SMPL_BONE_ORDER_NAMES
    model = load_asset_model()
    xpos, xquat, _, _ = kinematics_np(model, model.qpos0)
    verts, owner = [], []
    for g in range(model.ngeom):
        if model.geom_type[g] != 7:
            continue
        b = model.geom_bodyid[g]
        v = model.mesh_vert[model.geom_vertadr[g]:model.geom_vertadr[g] + model.geom_vertnum[g]] @ quat_to_mat(xquat[b]).T + xpos[b]
        verts.append(v)
        owner += [SMPL_BONE_ORDER_NAMES.index(model.body_names[b])] * len(v)
    verts0, owner = np.concatenate(verts), np.array(owner)
    joints0 = np.stack([xpos[model.body_names.index(n)] for n in SMPL_BONE_ORDER_NAMES])
    small = [SMPL_BONE_ORDER_NAMES.index(n) for n in ("L_Hand", "R_Hand", "L_Toe", "R_Toe")]

    def provider(betas, gender):
        b = np.r_[np.asarray(betas, dtype=np.float64).reshape(-1), np.zeros(3)]
        stature = 1.0 + 0.04 * np.tanh(b[0])
        girth = 1.0 + 0.08 * np.tanh(b[1]) + 0.02 * (int(gender) - 1)
        joints = joints0[0] + (joints0 - joints0[0]) * stature
        v = joints[owner] + (verts0 - joints0[owner]) * girth * stature
        keep = np.ones(len(v), dtype=bool)
        if vary_hulls:
            k = int(abs(b[2]) * 4) % 4  # 0..3 vertices dropped per small hull
            for j in small:
                idx = np.nonzero(owner == j)[0]
                # the vertices closest to the hull's centre line go first: the hull keeps its extent
                d = np.linalg.norm(v[idx] - v[idx].mean(0), axis=1)
                keep[idx[np.argsort(d)[:k]]] = False
        W = np.zeros((int(keep.sum()), 24))
        W[np.arange(int(keep.sum())), owner[keep]] = 1
        return v[keep], joints, W

    return provider
