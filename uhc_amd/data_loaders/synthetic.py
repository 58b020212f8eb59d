"""Synthetic AMASS-format clips (the real AMASS pickles and SMPL files are licensed and absent):
the shipped standing pose with smooth seeded joint-space perturbations, random heading, 150-300 frames
(SURVEY.md 8d, config 2).  Output matches what DatasetAMASSSingle consumes."""
import os

import numpy as np
from scipy.spatial.transform import Rotation as sRot

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def make_synthetic_amass(n_clips=64, seed=1, t_range=(150, 300), amp=0.3, root_height=0.91437225, model_root_z=0.0282):
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
    return out
