"""Expert features of a reference clip: ``Humanoid.qpos_fk`` (mirror of
uhc/smpllib/torch_smpl_humanoid.py:20-42,155-362).  Runs on whatever device `qpos` lives on, so the
clip bank can be built directly in HBM."""
from __future__ import annotations

import numpy as np
import torch

from ..utils.torch_utils import (get_angvel_fd_batch, get_qvel_fd_batch, quat_mul_vec_batch, quaternion_from_euler_rzyx,
                                 quaternion_multiply_batch, transform_vec_batch)
from .smpl_mujoco import SMPL_EE_NAMES


class Humanoid:
    def __init__(self, model_file=None, model=None):
        if model is None:
            raise ValueError("pass a compiled model (uhc_amd.model.mjcf.Model)")
        self.update_model(model)

    def update_model(self, model):
        self.model = model
        self.body_name = list(model.body_names[1:])
        n = len(self.body_name)
        self._offsets = torch.tensor(np.asarray(model.body_pos)[1:n + 1], dtype=torch.float64)
        self._i_offsets = torch.tensor(np.asarray(model.body_ipos)[1:n + 1], dtype=torch.float64)
        parents = np.asarray(model.body_parentid)[1:n + 1] - 1
        parents[0] = -1
        self._parents = parents
        self._ee_idx = [model.body_names.index(nm) - 1 for nm in SMPL_EE_NAMES]

    def get_head_idx(self):
        return self.model.body_names.index("Head") - 1

    def forward_kinematics_batch(self, rotations, root_rotations, root_positions):
        """rotations (B,J-1,4) local joint quats; returns world joint positions, body COMs, world quats (B,J,*)."""
        off = self._offsets.to(root_positions)
        ioff = self._i_offsets.to(root_positions)
        pos, com, rot = [], [], []
        for i in range(off.shape[0]):
            if self._parents[i] == -1:
                p, q = root_positions, root_rotations
            else:
                pq = rot[self._parents[i]]
                p = quat_mul_vec_batch(pq, off[i].expand_as(root_positions)) + pos[self._parents[i]]
                q = quaternion_multiply_batch(pq, rotations[:, i - 1, :])
            pos.append(p)
            rot.append(q)
            com.append(quat_mul_vec_batch(q, ioff[i].expand_as(root_positions)) + p)
        return torch.stack(pos, 1), torch.stack(com, 1), torch.stack(rot, 1)

    def qpos_fk(self, qpos, to_numpy=True):
        """qpos (T,76) -> the expert feature dict of torch_smpl_humanoid.py:234-261."""
        qpos = qpos.clone()
        T = qpos.shape[0]
        root_pos, root_rot, ang = qpos[:, :3], qpos[:, 3:7], qpos[:, 7:].reshape(T, -1, 3)
        J = ang.shape[1]
        body_quats = quaternion_from_euler_rzyx(ang[..., 0], ang[..., 1], ang[..., 2])
        wbpos, body_com, wbquat = self.forward_kinematics_batch(body_quats, root_rot, root_pos)
        bquat_full = torch.cat((root_rot[:, None, :], body_quats), dim=1)
        if T > 1:
            qvel = get_qvel_fd_batch(qpos[:-1], qpos[1:], 1 / 30)
            bangvel = get_angvel_fd_batch(bquat_full[:-1], bquat_full[1:], 1 / 30)
        else:
            qvel = torch.zeros((0, 6 + 3 * J)).to(qpos)
            bangvel = torch.zeros((0, J + 1, 3)).to(qpos)
        qvel = torch.cat((qvel[0:1], qvel), dim=0).clip(-10.0, 10.0)
        bangvel = torch.cat((bangvel[0:1], bangvel), dim=0)
        ee_w = wbpos[:, self._ee_idx, :]
        ee_l = transform_vec_batch((ee_w - wbpos[:, 0:1, :]).reshape(-1, 3), root_rot.repeat_interleave(len(self._ee_idx), 0)).reshape(T, -1, 3)
        out = {
            "qpos": qpos, "qvel": qvel, "wbpos": wbpos.reshape(T, -1), "wbquat": wbquat.reshape(T, -1),
            "bquat": bquat_full.reshape(T, -1), "body_com": body_com.reshape(T, -1), "rlinv": qvel[:, :3].clone(),
            "rlinv_local": transform_vec_batch(qvel[:, :3], qpos[:, 3:7]), "rangv": qvel[:, 3:6].clone(),
            "bangvel": bangvel.reshape(T, -1), "ee_wpos": ee_w.reshape(T, -1), "ee_pos": ee_l.reshape(T, -1),
            "com": body_com[:, 0].reshape(T, -1),
        }
        out["height_lb"] = out["qpos"][:, 2].min()
        if to_numpy:
            out = {k: v.cpu().numpy() for k, v in out.items()}
        out["len"] = T
        return out
