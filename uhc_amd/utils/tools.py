"""`uhc.utils.tools` of the reference (uhc/utils/tools.py): the checkpoint unpickler and the simulator-driven expert features.

* `CustomUnpickler` (:7-18): reference checkpoints pickle `ZFilter` / `RunningStat` under whatever module path their authors' tree had at the
  time; both names resolve to this build's classes.
* `get_expert` (:21-101): the expert feature dictionary of a clip computed by stepping the ENV through the clip's poses -- `set_state`,
  `sim.forward()`, read the body poses -- instead of the torch forward kinematics `load_expert` uses (`Humanoid.qpos_fk`,
  uhc_amd/smpllib/torch_smpl_humanoid.py).  Same keys and shapes; velocities by `get_qvel_fd_new` at env.dt, clipped to +-10.
* `get_expert_master` (:104-...) converts the pose to the masterfoot model's joint set first; that model class is not built."""
import pickle
from collections import defaultdict

import numpy as np

from .math_utils import de_heading, get_angvel_fd, get_heading_q, get_qvel_fd_new, transform_vec


class CustomUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if name in ("ZFilter", "RunningStat"):
            from ..khrylib.utils import zfilter
            return getattr(zfilter, name)
        return super().find_class(module, name)


def get_expert(expert_qpos, expert_meta, env):
    """env: the single-env facade (HumanoidEnv).  Its state is restored afterwards."""
    expert_qpos = np.asarray(expert_qpos, dtype=np.float64)
    nq = env.vec.qpos_lim
    saved = (env.data.qpos.copy(), env.data.qvel.copy())
    rec = defaultdict(list)
    head = env.model._body_name2id["Head"]
    for i, qpos in enumerate(expert_qpos):
        q = saved[0].copy()
        q[:nq] = qpos[:nq]
        env.set_state(q, np.zeros_like(saved[1]))  # = data.qpos[:76] = qpos; sim.forward()
        d = env.data
        rec["rq_rmh"].append(de_heading(qpos[3:7]))
        rec["ee_pos"].append(env.get_ee_pos(env.cc_cfg.obs_coord))
        rec["ee_wpos"].append(env.get_ee_pos(None))
        rec["wbpos"].append(env.get_wbody_pos())
        rec["wbquat"].append(env.get_wbody_quat())
        rec["bquat"].append(env.get_body_quat())
        rec["com"].append(np.array(env.get_com()))
        rec["body_com"].append(env.get_body_com())
        rec["head_pose"].append(np.concatenate((d.body_xpos[head], d.body_xquat[head])))
        if i > 0:
            qvel = get_qvel_fd_new(expert_qpos[i - 1], qpos, env.dt).clip(-10.0, 10.0)
            rec["qvel"].append(qvel)
            rec["rlinv"].append(qvel[:3].copy())
            rec["rlinv_local"].append(transform_vec(qvel[:3].copy(), qpos[3:7], env.cc_cfg.obs_coord))
            rec["rangv"].append(qvel[3:6].copy())
            rec["bangvel"].append(get_angvel_fd(rec["bquat"][i - 1], rec["bquat"][i], env.dt))
    for k in ("qvel", "rlinv", "rlinv_local", "rangv", "bangvel"):  # frame 0 repeats frame 1 (:86-94)
        rec[k].insert(0, rec[k][0].copy())
    expert = {k: np.vstack(v) for k, v in rec.items()}
    expert["qpos"], expert["meta"], expert["len"] = expert_qpos, expert_meta, expert_qpos.shape[0]
    expert["height_lb"] = expert_qpos[:, 2].min()
    expert["head_height_lb"] = expert["head_pose"][:, 2].min()
    if expert_meta.get("cyclic", False):
        expert["init_heading"] = get_heading_q(expert_qpos[0, 3:7])
        expert["init_pos"] = expert_qpos[0, :3].copy()
    env.set_state(*saved)
    return expert


def get_expert_master(expert_qpos, expert_meta, env):
    raise NotImplementedError("the masterfoot model class (uhc/smpllib/smpl_robot.py:1336-1414: 49 bodies, capsule geoms) is not built: DESIGN.md section 0, row (f)-4")
