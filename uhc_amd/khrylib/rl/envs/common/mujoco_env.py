"""`MujocoEnv` -- the base class the reference's envs derive from (uhc/khrylib/rl/envs/common/mujoco_env.py:12-176), over this build's
batched simulator instead of mujoco-py: the single-environment facade `uhc_amd.envs.humanoid_im.HumanoidEnv` derives from it.

What the base owns in the reference and here: the control-step length (`dt` = model timestep x frame_skip, :116), the seeded
`np_random` (:66-70), `set_state` = write qpos / qvel + `sim.forward()` (:106-113), `state_vector` (:168), the body-frame helpers
(:165-176), `reset` -> `reset_model` (:95-104).  Rendering (:85-93, :124-162) is outside this build's scope (the viewer is
not on the hot path) and raises."""
import numpy as np
import torch


class MujocoEnv:
    frame_skip = 15

    # -- what a subclass provides: `self.vec` (a VecHumanoidEnv of one env) and `self.data` (host snapshot of the sim fields)
    @property
    def dt(self):
        return self.vec.model.timestep * self.frame_skip

    @dt.setter
    def dt(self, _):  # (the facade binds the batched env's own value; kept so that `self.dt = ...` in subclasses stays legal)
        pass

    def seed(self, seed=None):
        out = self.vec.seed(seed)
        self.np_random = self.vec.np_random
        return out

    def reset_model(self):
        raise NotImplementedError("reset_model is the subclass's (mujoco_env.py:78-83)")

    def reset(self):
        return self.reset_model()

    def set_state(self, qpos, qvel):
        """mujoco_env.py:106-113: the state is written and `sim.forward()` run (here: uhc_batch_set_state, which does both)."""
        q = np.asarray(qpos, dtype=np.float64).reshape(1, -1)
        v = np.asarray(qvel, dtype=np.float64).reshape(1, -1)
        assert q.shape[1] == self.vec.model.nq and v.shape[1] == self.vec.model.nv
        self.vec.sim.set_state(torch.from_numpy(q), torch.from_numpy(v), torch.zeros(1, dtype=torch.int32))

    def state_vector(self):
        d = self.data
        return np.concatenate([d.qpos, d.qvel])

    def get_body_com(self, body_name):
        return self.data.get_body_xpos(body_name)

    def vec_body2world(self, body_name, vec):
        from ....utils.transformation import quaternion_matrix
        d = self.data
        R = quaternion_matrix(d.body_xquat[d._names.index(body_name)])[:3, :3]
        return R @ np.asarray(vec, dtype=np.float64)

    def pos_body2world(self, body_name, pos):
        return self.vec_body2world(body_name, pos) + self.data.get_body_xpos(body_name)

    def render(self, *a, **k):
        raise RuntimeError("rendering needs mujoco-py's viewer, which this build replaces with nothing (SURVEY 8: out of scope)")

    def close(self):
        self.vec.close()
