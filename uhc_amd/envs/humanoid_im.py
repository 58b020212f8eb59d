"""Humanoid imitation environments.

``VecHumanoidEnv`` -- n_env environments stepped together on one MI355X through the C-ABI
(uhc_env_* / uhc_batch_*); observations, rewards and flags stay in HBM.
``HumanoidEnv``    -- the reference's single-env gym-like surface (uhc/envs/humanoid_im.py:47-1465:
``reset, step, load_expert, set_mode, seed, observation_space, action_space, data, model, ...``) as a
1-env view of the same machinery, so reference-style callers (AgentCopycat, eval scripts) run unmodified.
"""
from __future__ import annotations

import dataclasses
import types

import numpy as np
import torch

from .. import sim as S
from .._capi import REWARD_IDS, REWARD_PARTS, env_desc
from ..khrylib.rl.envs.common.mujoco_env import MujocoEnv
from ..smpllib.smpl_mujoco import SMPLConverter, smpl_to_qpose
from ..smpllib.torch_smpl_humanoid import Humanoid


class _Box:
    def __init__(self, dim):
        self.shape = (dim,)
        self.low, self.high = -np.ones(dim), np.ones(dim)


class VecHumanoidEnv:
    def __init__(self, cfg, n_env, device=0, mode="train", model=None, shape_models=None, objects=None):
        """shape_models: optional list of further models with the topology of `model` (body shapes, cf. the smpl_shape configs:
        the reference rebuilds the model from each clip's beta); `set_clip_bank(..., clip_model=)` maps clips onto them.
        objects: free objects behind the humanoid (the reference appends them in the generator from expert["obj_info"], a body with a free
        joint and a mesh geom each, contype = conaffinity = 1, friction 1: uhc/smpllib/smpl_robot.py:1200-1252): dict(hulls=[(ntri, 3, 3)
        triangle soup per object], density=1000.0, friction=1.0, condim=3) or just the list of hulls.  Every clip of the bank then carries
        `obj_pose` (T, 7 K): a reset puts the objects there (reset_model, humanoid_im.py:1284-1287); observation, reward and termination keep
        reading the humanoid alone (qpos[:qpos_lim], :421-422)."""
        self.cc_cfg = self.cfg = cfg
        self.mode = mode
        self.n_env = int(n_env)
        if getattr(cfg, "masterfoot", False) or dict(cfg.robot_cfg or {}).get("masterfoot", False):
            # uhc/envs/humanoid_im.py:54-58: Robot(cfg.robot_cfg, masterfoot=cfg.masterfoot).  The generator builds that model (uhc_amd/smpllib/smpl_robot.py:
            # Robot({"masterfoot": True}): 49 bodies / 147 dofs, capsule toes) and the CPU checker steps it; the HIP step kernels hold 128 dofs
            raise NotImplementedError("masterfoot: the 147-dof model (12 capsule bodies of three hinges under each ankle) is generated and compiled -- "
                                      "uhc_amd.smpllib.smpl_robot.Robot({'mesh': True, 'masterfoot': True}) -- but the HIP step kernels hold nv <= 128: not runnable on the device")
        if model is None:
            # the reference never loads the asset file: its env model comes out of Robot(cfg.robot_cfg) (humanoid_im.py:52-64) -- body-body
            # collisions on, rel_joint_lm ranges, ball joints if asked.  Without the licensed SMPL files the shipped asset stands in for
            # the generator's body, put into the form the generator emits; `model=` / `shape_models=` take generated models as they are
            from ..smpllib.smpl_robot import robot_variant
            model = robot_variant(S.load_asset_model(getattr(cfg, "mujoco_model", "humanoid_smpl_neutral_mesh")), cfg.robot_cfg)
            shape_models = [robot_variant(m, cfg.robot_cfg) for m in (shape_models or [])]  # shapes of the asset: converted with it
        from ..model.mjcf import add_free_bodies, trailing_free_bodies
        if trailing_free_bodies(model):
            raise ValueError("pass the humanoid's model and the objects apart (objects=): the env needs the humanoid alone for its kinematics")
        # robot.ball (config/copycat_ball): one ball joint per bone, nq = 99 -- `use_quat` in the reference (humanoid_im.py:52).  Gains, limits
        # and the expert's forward kinematics are read off each model itself (body names and offsets: the same in both joint layouts), as
        # the reference builds converter and Humanoid from the model the clip's Robot(beta) produced (:104-124, :192)
        self.use_quat = int(model.nq) != int(model.nv) + 1
        if self.use_quat:
            if int(model.nq) != 7 + 4 * (int(model.nbody) - 2) or int(model.nv) != 6 + 3 * (int(model.nbody) - 2):
                raise NotImplementedError("robot.ball: a free root and one ball joint per further body is what the generator emits")
            if cfg.action_type != "torque" or cfg.residual_force or cfg.meta_pd or cfg.meta_pd_joint:
                raise NotImplementedError("robot.ball runs with action_type torque and no residual force / meta-PD (config/copycat_ball/*.yml); "
                                          "the stable-PD controller is written for scalar joints")
            if cfg.obs_v != 2 or cfg.reward_id not in ("world_rfc_implicit_quat", "world_rfc_implicit"):
                raise NotImplementedError("robot.ball: obs_v 2 (get_full_obs_v2_quat) and reward world_rfc_implicit_quat are built")
        iters = int(getattr(cfg, "pgs_iterations", 300))
        self.body_models = [dataclasses.replace(m, iterations=iters, solver=int(getattr(cfg, "contact_solver", 0))) for m in [model] + list(shape_models or [])]
        self.body_model = self.body_models[0]  # the humanoid alone: kinematics of the expert, gains, body names
        self.num_obj, self.objects = 0, None
        if objects is not None:
            spec = dict(objects) if isinstance(objects, dict) else dict(hulls=list(objects))
            hulls = [np.asarray(h, dtype=np.float64) for h in spec["hulls"]]
            self.num_obj = len(hulls)
            self.objects = dict(hulls=hulls, density=float(spec.get("density", 1000.0)), friction=spec.get("friction", 1.0), condim=int(spec.get("condim", 3)))
        if self.num_obj:
            # (qpos0 of the objects: side by side above the floor, overwritten by every reset's obj_pose)
            park = np.stack([np.r_[2.0 + 1.0 * k, 2.0, 1.0, 1.0, 0.0, 0.0, 0.0] for k in range(self.num_obj)])
            self.models = [add_free_bodies(m, self.objects["hulls"], park, density=self.objects["density"], friction=self.objects["friction"],
                                           condim=self.objects["condim"]) for m in self.body_models]
        else:
            self.models = list(self.body_models)
        self.model = self.models[0]
        self.qpos_lim, self.qvel_lim, self.body_lim = int(self.body_model.nq), int(self.body_model.nv), int(self.body_model.nbody)  # humanoid_im.py:113-115
        self.base_rot = cfg.data_specs.get("base_rot", [0.7071, 0.7071, 0.0, 0.0])
        self.rfc_rate = 1 if not cfg.rfc_decay else 0
        kin = self.body_model
        self.converter = SMPLConverter(kin, kin, smpl_model=cfg.robot_cfg.get("model", "smpl"))
        self.ctrl = S.make_ctrl(kin, meta_pd=cfg.meta_pd, meta_pd_joint=cfg.meta_pd_joint, residual_force=cfg.residual_force,
                                residual_force_mode=cfg.residual_force_mode, residual_force_scale=cfg.residual_force_scale,
                                residual_force_lim=cfg.residual_force_lim, rfc_rate=self.rfc_rate, action_type=cfg.action_type,
                                pd_mul=cfg.get("pd_mul", 1), tq_mul=cfg.get("tq_mul", 1), base_rot=self.base_rot,
                                residual_force_bodies=cfg.residual_force_bodies, residual_force_torque=cfg.residual_force_torque)
        self.sim = S.SimBatch(self.models, self.ctrl, self.n_env, device=device)
        import os
        self.sim.set_kernel_path(int(os.environ.get("UHC_KERNEL_PATH", "2")))  # 2 = sticky tiers: every env starts a step in the tier that computed its last one
        self.device = self.sim.device
        thresh = cfg.get("body_diff_thresh", 0.5) if mode == "train" else cfg.get("body_diff_thresh_test", 0.5)
        if cfg.env_term_body not in ("body", "root"):
            raise NotImplementedError("env_term_body 'body' and 'root' are built ('Head' reads expert['head_height_lb'], which the reference never sets)")
        reward_v = REWARD_IDS.get(cfg.reward_id)
        if reward_v is None:
            raise NotImplementedError(f"reward_id {cfg.reward_id!r}: built are {sorted(REWARD_IDS)} (the local_rfc_* rewards are not)")
        # (obs_v 4, humanoid_im.py:769-861, returns a (obs_full, local_obs, global_obs) tuple in the reference, which its own observation_space / ZFilter cannot
        #  consume; here the env's observation is obs_full -- global block | shape | one row of 26 per non-root body -- and HumanoidEnv.get_full_obs_v4 hands out the tuple)
        if cfg.obs_v == 4 and cfg.get("robot_cfg", {}).get("ball", False):
            raise NotImplementedError("obs_v 4 reads three hinge angles per body: not defined for the ball-joint humanoid")
        if cfg.obs_coord != "root" or (cfg.obs_v != 0 and cfg.obs_vel != "full"):
            raise NotImplementedError("obs_coord 'root' (every reference config) and, for obs_v >= 1, obs_vel 'full' are built")
        self.n_reward_parts = REWARD_PARTS[reward_v]
        self.env = S.EnvBatch(self.sim, env_desc(kin, obs_v=cfg.obs_v, reward_v=reward_v, obs_heading=cfg.obs_heading, root_deheading=cfg.root_deheading,
                                                 obs_phase=cfg.obs_phase, obs_vel=cfg.obs_vel, env_term_body=cfg.env_term_body, fut_frames=cfg.get("fut_frames", 10), fut_skip=cfg.get("skip", 10), has_shape=cfg.has_shape and cfg.get("has_shape_obs", True),
                                                 env_episode_len=cfg.env_episode_len, env_expert_trail_steps=cfg.env_expert_trail_steps,
                                                 body_diff_thresh=thresh, reward_weights=cfg.reward_weights,
                                                 jpos_diffw=self.converter.get_new_diff_weight(), num_obj=self.num_obj))
        self.ndof = self.model.nu
        self.body_vf_dim = self.ctrl.body_vf_dim
        self.vf_dim = 0 if not cfg.residual_force else (6 if cfg.residual_force_mode == "implicit" else self.ctrl.n_vf_body * self.body_vf_dim)
        self.meta_pd_dim = 30 if cfg.meta_pd else (2 * self.ndof if cfg.meta_pd_joint else 0)
        self.action_dim = self.ctrl.action_dim
        self.obs_dim = self.env.obs_dim
        self.observation_space, self.action_space = _Box(self.obs_dim), _Box(self.action_dim)
        self.humanoid = Humanoid(model=kin)
        self._humanoids = {0: self.humanoid}
        self.np_random = np.random.RandomState()
        self._end_reward = 0.0
        self.dt = self.model.timestep * 15
        self.clip_keys, self._clip_index, self._clip_len = [], {}, None

    # ---- expert clips -------------------------------------------------------------------------------------
    def expert_features(self, sample, model_index=0):
        """load_expert's feature computation (humanoid_im.py:182-215): AMASS window -> qpos -> qpos_fk, on the clip's OWN model (the
        reference rebuilds model, converter and Humanoid from the clip's Robot(beta) before it computes them)."""
        kin = self.body_models[model_index]
        if model_index not in self._humanoids:
            self._humanoids[model_index] = Humanoid(model=kin)
        kw = dict(pose=sample["pose_aa"], trans=np.asarray(sample["trans"]).squeeze(), model=self.cc_cfg.robot_cfg.get("model", "smpl"),
                  count_offset=self.cc_cfg.robot_cfg.get("mesh", True))
        feat = self._humanoids[model_index].qpos_fk(torch.from_numpy(smpl_to_qpose(mj_model=kin, **kw)))
        if self.use_quat:
            # load_expert computes the quaternion pose too (humanoid_im.py:193-200).  Its joint quaternions ARE the expert's local body
            # quaternions (bquat); its root quaternion comes out of another conversion routine than the Euler pose's (1.5e-3 rad apart):
            # the frame record carries it in the root slot of qpos, the joint angles behind it stay (unused by a torque-driven env)
            feat["qpos_quat"] = smpl_to_qpose(mj_model=kin, use_quat=True, **kw)
            q = np.array(feat["qpos"], dtype=np.float64, copy=True)
            q[:, 3:7] = feat["qpos_quat"][:, 3:7]
            feat["qpos_record"] = q
        return feat

    def _clip_obj_pose(self, c, T):
        """expert["obj_pose"] of a clip as (T, 7 num_obj); the loader stores pose_aa under that key for clips without objects
        (dataset_amass_single.py:133, `has_obj` = the shapes differ, :246)."""
        op = c.get("obj_pose") if isinstance(c, dict) else None
        op = None if op is None else np.asarray(op, dtype=np.float64)
        if op is None or op.ndim != 2 or op.shape != (T, 7 * self.num_obj):
            raise ValueError(f"the env carries {self.num_obj} objects: every clip needs obj_pose of shape (T, {7 * self.num_obj})")
        return op

    def set_clip_bank(self, clips: dict, clip_model: dict = None):
        """clips: {key: sample dict with pose_aa/trans/beta/gender (+ obj_pose when the env carries objects) of the WHOLE clip}.  Builds the
        HBM bank once.  clip_model: {key: index into [model] + shape_models}: the body shape every episode of that clip runs on."""
        frames, starts, betas, lens, objs = [], [], [], [], []
        n = 0
        self.clip_keys = list(clips.keys())
        cm = [int(clip_model[k]) if clip_model else 0 for k in self.clip_keys]
        for k, mi in zip(self.clip_keys, cm):
            c = clips[k]
            ft = self.expert_features(c, mi)
            fr = S.pack_expert_frames(dict(ft, qpos=ft["qpos_record"]) if self.use_quat else ft)
            frames.append(fr)
            if self.num_obj:
                objs.append(self._clip_obj_pose(c, fr.shape[0]))
            starts.append(n)
            lens.append(fr.shape[0])
            n += fr.shape[0]
            beta = np.asarray(c["beta"], dtype=np.float64)
            beta = beta[0] if beta.ndim == 2 else beta
            beta = np.concatenate([beta, np.zeros(16 - beta.shape[0])]) if beta.shape[0] < 16 else beta[:16]
            g = np.asarray(c["gender"]).reshape(-1)[0]
            betas.append(np.r_[beta, float(g)])
        self._clip_index = {k: i for i, k in enumerate(self.clip_keys)}
        self._clip_len = np.array(lens)
        self.env.set_bank(torch.from_numpy(np.concatenate(frames)), torch.tensor(starts, dtype=torch.int32), torch.from_numpy(np.stack(betas)))
        self.env.set_obj_pose(torch.from_numpy(np.concatenate(objs)) if self.num_obj else None)
        self.env.set_clip_models(torch.tensor(cm, dtype=torch.int32) if clip_model else None)

    def set_clip_bank_from_loader(self, data_loader, clip_model=None):
        clips = {k: dict(pose_aa=data_loader.data["pose_aa"][k], trans=data_loader.data["trans"][k], beta=data_loader.data["beta"][k],
                         gender=data_loader.data["gender"][k], obj_pose=data_loader.data.get("obj_pose", {}).get(k)) for k in data_loader.data_keys}
        self.set_clip_bank(clips, clip_model=clip_model)

    def _window_len(self, fr_start, fr_end):
        n = np.asarray(fr_end) - np.asarray(fr_start)
        if self.cc_cfg.obs_v == 3:  # load_expert shortens the episode by 30 frames for the look-ahead observation (humanoid_im.py:214-215)
            n = np.maximum(n - 30, 1)
        return n

    def assign(self, env_ids, keys, fr_start, fr_end):
        """load_expert bookkeeping for a set of envs: env i imitates frames [fr_start, fr_end) of clip keys[i]."""
        cid = torch.tensor([self._clip_index[k] for k in keys], dtype=torch.int32)
        fs = torch.as_tensor(np.asarray(fr_start), dtype=torch.int32)
        fl = torch.as_tensor(self._window_len(fr_start, fr_end), dtype=torch.int32)
        self.env.assign(torch.as_tensor(env_ids, dtype=torch.int32), cid, fs, fl)

    # ---- gym-like batched surface -----------------------------------------------------------------------
    def set_mode(self, mode):
        self.mode = mode

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def set_rfc_rate(self, rate):
        self.rfc_rate = rate
        if self.ctrl.rfc_mode != 2:  # rfc_explicit scales by residual_force_scale alone (humanoid_im.py:1110-1111)
            self.sim.set_rfc_scale(self.cc_cfg.residual_force_scale * rate)

    def reset(self, env_ids=None):
        """reset_model (humanoid_im.py:1245-1299) on the listed envs; returns the obs tensor view (n_env, obs_dim)."""
        ids = torch.arange(self.n_env, dtype=torch.int32) if env_ids is None else torch.as_tensor(env_ids, dtype=torch.int32)
        noise = None
        if self.mode == "train" and self.cc_cfg.env_init_noise > 0 and not self.use_quat:  # (noise is defined on joint ANGLES, :1262-1266)
            noise = torch.from_numpy(self.np_random.normal(loc=0.0, scale=self.cc_cfg.env_init_noise, size=(ids.shape[0], self.ndof)))
        self.env.reset(ids, noise)
        return self.obs

    def set_next(self, env_ids, keys, fr_start, fr_end):
        """Queue the next episode's window for the listed envs (load_expert + reset_model happen on the device at `done`)."""
        ids = np.asarray(env_ids, dtype=np.int32)
        cid = np.fromiter((self._clip_index[k] for k in keys), dtype=np.int32, count=len(ids))
        fs = np.asarray(fr_start, dtype=np.int32)
        fl = np.asarray(self._window_len(fr_start, fr_end), dtype=np.int32)
        noise = None
        if self.mode == "train" and self.cc_cfg.env_init_noise > 0 and not self.use_quat:
            noise = self.np_random.normal(loc=0.0, scale=self.cc_cfg.env_init_noise, size=(ids.shape[0], self.ndof))
        self.env.set_next_host(ids, cid, fs, fl, noise)  # one asynchronous copy: the sampling loop must not wait for the GPU here

    def auto_reset(self):
        self.env.auto_reset()

    @property
    def end_reward(self):
        return self._end_reward

    @end_reward.setter
    def end_reward(self, v):  # the library adds it to the episode return of a step that ends its clip
        self._end_reward = float(v)
        self.env.set_end_reward(v)

    def step(self, action, active=None):
        self.env.step(action, active)
        return self.obs, self.reward, self.done, {"fail": self.env.field(S.E_FAIL), "end": self.env.field(S.E_END), "percent": self.env.field(S.E_PERCENT)}

    obs = property(lambda self: self.env.field(S.E_OBS))
    reward = property(lambda self: self.env.field(S.E_REWARD))
    reward_parts = property(lambda self: self.env.field(S.E_REWARD_PARTS)[:, :self.n_reward_parts])
    done = property(lambda self: self.env.field(S.E_DONE))
    cur_t = property(lambda self: self.env.field(S.E_CUR_T))

    def close(self):
        self.env.close()
        self.sim.close()


class _Data:
    """Host snapshot of mujoco-py's `sim.data` fields the reference reads."""

    def __init__(self, venv):
        s = venv.sim
        nb = venv.model.nbody
        self.qpos = s.field(S.F_QPOS)[0].cpu().numpy()
        self.qvel = s.field(S.F_QVEL)[0].cpu().numpy()
        self.body_xpos = s.field(S.F_XPOS)[0].cpu().numpy().reshape(nb, 3)
        self.body_xquat = s.field(S.F_XQUAT)[0].cpu().numpy().reshape(nb, 4)
        self.xipos = s.field(S.F_XIPOS)[0].cpu().numpy().reshape(nb, 3)
        self.ctrl = s.field(S.F_CTRL)[0].cpu().numpy()
        self.qfrc_applied = s.field(S.F_QFRC_APPLIED)[0].cpu().numpy()
        self.qfrc_bias = s.field(S.F_QFRC_BIAS)[0].cpu().numpy()
        self.ncon = int(s.field(S.F_NCON)[0].item())
        self._names = venv.model.body_names

    def get_body_xipos(self, name):
        return self.xipos[self._names.index(name)]

    def get_body_xpos(self, name):
        return self.body_xpos[self._names.index(name)]


class HumanoidEnv(MujocoEnv):
    """Single-environment facade with the reference's constructor and methods (humanoid_im.py:49-70); like the reference's class it derives
    from `MujocoEnv` (seed, dt, set_state + forward, state_vector, body-frame helpers: uhc_amd/khrylib/rl/envs/common/mujoco_env.py)."""

    def __init__(self, cfg, init_expert, data_specs, mode="train", no_root=False, device=0, body_provider=None):
        """body_provider: (betas, gender) -> (vertices, joints, skin weights) for the shape -> model generator; default: the SMPL files
        under <base_dir>/data/smpl when present, else the shipped asset in generated form stands for every body (robot_variant)."""
        from ..smpllib.smpl_robot import Robot, default_body_provider
        self.cc_cfg, self.mode, self.no_root, self._device = cfg, mode, no_root, device
        provider = body_provider or default_body_provider(cfg)
        self.smpl_robot = Robot(cfg.robot_cfg, body_provider=provider) if provider is not None else None  # humanoid_im.py:53-58
        self._robot_tag = None
        self._obj_tag, objects = self._objects_of(init_expert)
        self.vec = VecHumanoidEnv(cfg, 1, device=device, mode=mode, model=self._robot_model(init_expert), objects=objects)
        self._bind()
        self.body_diffw = self.vec.converter.get_new_diff_weight()[1:]
        self.jpos_diffw = self.vec.converter.get_new_diff_weight()[:, None]
        self.end_reward, self.start_ind, self.expert = 0.0, 0, None
        self.prev_bquat = None
        self.load_expert(init_expert, reload_robot=False)  # (the model of the first clip is already in place)

    def _robot_model(self, expert_data):
        """reset_robot (humanoid_im.py:154-180): the model generated from the clip's beta and gender, or None without a generator."""
        if self.smpl_robot is None:
            return None
        beta = np.asarray(expert_data["beta"], dtype=np.float64)
        beta = beta[0] if beta.ndim == 2 else beta
        g = int(np.asarray(expert_data["gender"]).reshape(-1)[0])
        self._robot_tag = (beta.round(6).tobytes(), g)
        cache = self.__dict__.setdefault("_robot_models", {})
        if self._robot_tag not in cache:
            cache[self._robot_tag] = self.smpl_robot.load_from_skeleton(beta, gender=[g]).get_model()
        return cache[self._robot_tag]

    @staticmethod
    def _objects_of(expert_data):
        """reset_robot hands expert["obj_info"] to the generator, which appends one free body + mesh geom per entry (humanoid_im.py:154-170,
        smpl_robot.py:1200-1252: mesh files named after the GRAB objects).  Here: expert["obj_mesh"] = list of (ntri, 3, 3) triangle soups, or
        obj_info = paths of binary STL files.  Returns (tag, objects spec or None)."""
        if not expert_data.get("has_obj", False) or int(expert_data.get("num_obj", 0)) == 0:
            return None, None
        if expert_data.get("obj_mesh") is not None:
            hulls = [np.asarray(h, dtype=np.float64) for h in expert_data["obj_mesh"]]
        elif expert_data.get("obj_info") is not None:
            from ..model.mjcf import load_binary_stl
            hulls = [load_binary_stl(str(f)) for f in np.asarray(expert_data["obj_info"]).reshape(-1)]
        else:
            raise ValueError("a clip with objects needs their meshes: expert['obj_mesh'] (triangle soups) or expert['obj_info'] (STL paths)")
        if len(hulls) != int(expert_data["num_obj"]):
            raise ValueError(f"{len(hulls)} object meshes for num_obj = {expert_data['num_obj']}")
        tag = tuple(h.round(9).tobytes() for h in hulls)
        return tag, dict(hulls=hulls, density=float(expert_data.get("obj_density", 1000.0)))

    def _bind(self):
        v = self.vec
        self.model, self.converter, self.humanoid = v.model, v.converter, v.humanoid
        self.observation_space, self.action_space = v.observation_space, v.action_space
        self.ndof, self.vf_dim, self.meta_pd_dim, self.action_dim, self.obs_dim = v.ndof, v.vf_dim, v.meta_pd_dim, v.action_dim, v.obs_dim
        self.base_rot, self.dt, self.frame_skip = v.base_rot, v.dt, 15
        self.np_random = v.np_random

    rfc_rate = property(lambda self: self.vec.rfc_rate, lambda self, r: self.vec.set_rfc_rate(r))
    cur_t = property(lambda self: int(self.vec.cur_t[0].item()))
    data = property(lambda self: _Data(self.vec))

    def set_mode(self, mode):
        self.mode = mode
        self.vec.set_mode(mode)

    def load_expert(self, expert_data, reload_robot=True):
        if reload_robot:  # humanoid_im.py:189-190: a new body for a new beta / gender, new objects for a new obj_info
            tag, otag = self._robot_tag, self._obj_tag
            model = self._robot_model(expert_data) if self.smpl_robot is not None else None
            self._obj_tag, objects = self._objects_of(expert_data)
            if tag != self._robot_tag or otag != self._obj_tag:
                rate, seed_state = self.vec.rfc_rate, self.vec.np_random
                self.vec.close()
                self.vec = VecHumanoidEnv(self.cc_cfg, 1, device=self._device, mode=self.mode, model=model, objects=objects)
                self.vec.np_random = seed_state
                self.vec.set_rfc_rate(rate)
                self._bind()
        self.expert = dict(expert_data)
        self.expert["meta"] = {"cyclic": False, "seq_name": expert_data["seq_name"]}
        self.expert.update(self.vec.expert_features(expert_data))
        self.vec.set_clip_bank({expert_data["seq_name"]: expert_data})
        self.vec.assign([0], [expert_data["seq_name"]], [0], [self.expert["len"]])

    def reset_model(self):  # (MujocoEnv.reset calls it: mujoco_env.py:95-104)
        obs = self.vec.reset()[0].cpu().numpy()
        self.prev_bquat = self.get_body_quat()  # humanoid_im.py:1269 (reset_model -> get_body_quat)
        return obs

    def step(self, a):
        self.prev_bquat = self.get_body_quat()  # humanoid_im.py:1195-1196: the reward's finite difference reads the pre-step quaternions
        act = torch.as_tensor(np.asarray(a, dtype=np.float64).reshape(1, -1), device=self.vec.device)
        self.vec.step(act)
        info = {"fail": bool(self.vec.env.field(S.E_FAIL)[0].item()), "end": bool(self.vec.env.field(S.E_END)[0].item()),
                "percent": float(self.vec.env.field(S.E_PERCENT)[0].item())}
        self.last_reward = (float(self.vec.reward[0].item()), self.vec.reward_parts[0].cpu().numpy())
        return self.vec.obs[0].cpu().numpy(), 1.0, bool(self.vec.done[0].item()), info

    def get_full_obs_v4(self, delta_t=0):
        """humanoid_im.py:769-861: (obs_full, local_obs (23, 26), global_obs).  The device kernel writes obs_full for the current step (cfg.obs_v == 4);
        the two parts are cut out of it."""
        if self.cc_cfg.obs_v != 4 or delta_t != 0:
            raise NotImplementedError("get_full_obs_v4: the env computes it for cfg.obs_v == 4, delta_t == 0")
        full = self.vec.obs[0].cpu().numpy().copy()
        ng = full.shape[0] - 26 * (self.vec.body_lim - 2)
        return full, full[ng:].reshape(-1, 26).copy(), full[:ng].copy()

    def get_expert_index(self, t):
        return min(self.start_ind + t, self.expert["len"] - 1)

    def get_expert_attr(self, attr, ind):
        return self.expert[attr][ind].copy()

    def get_wbody_pos(self, selectList=None):
        d = self.data
        return d.body_xpos[1:self.vec.body_lim].copy().ravel() if selectList is None else np.concatenate([d.get_body_xpos(b) for b in selectList])

    def get_humanoid_qpos(self):
        return self.data.qpos.copy()[:self.vec.qpos_lim]  # humanoid_im.py:1417-1421

    def get_humanoid_qvel(self):
        return self.data.qvel.copy()[:self.vec.qvel_lim]

    def get_obj_qpos(self):
        return self.data.qpos.copy()[self.vec.qpos_lim:]  # humanoid_im.py:1423-1428

    def get_obj_qvel(self):
        return self.data.qvel.copy()[self.vec.qvel_lim:]

    def get_world_vf(self):
        return None

    # ---- accessors the reference's reward functions and eval loop call (humanoid_im.py:902-965) --------------------------------
    def get_expert_qpos(self, delta_t=0):
        return self.get_expert_attr("qpos", self.get_expert_index(self.cur_t + delta_t))

    def get_expert_qvel(self, delta_t=0):
        return self.get_expert_attr("qvel", self.get_expert_index(self.cur_t + delta_t))

    def fail_safe(self):
        """Teleport to the expert state of the current frame and refresh the kinematics (humanoid_im.py:902-905)."""
        ind = self.get_expert_index(self.cur_t)
        # (ball joints: the quaternion expert pose -- the model's own 99 coordinates -- not the hinge angles)
        q = self.get_expert_attr("qpos_quat" if self.vec.use_quat and "qpos_quat" in self.expert else "qpos", ind)
        v = self.get_expert_qvel()
        if self.vec.num_obj:  # data.qpos[:qpos_lim] = expert pose: the objects stay where they are
            q, v = np.r_[q, self.get_obj_qpos()], np.r_[v, self.get_obj_qvel()]
        self.set_state(q, v)  # (MujocoEnv: write the state, sim.forward())

    def get_head_idx(self):
        return self.model._body_name2id["Head"] - 1

    def get_ee_pos(self, transform):
        """World (transform None) or root / heading relative positions of SMPL_EE_NAMES (humanoid_im.py:910-923)."""
        from ..smpllib.smpl_mujoco import SMPL_EE_NAMES
        from ..utils.math_utils import transform_vec
        d = self.data
        root_pos, root_q = d.qpos[:3], d.qpos[3:7].copy()
        out = []
        for name in SMPL_EE_NAMES:
            v = d.get_body_xpos(name)
            if transform is not None:
                v = transform_vec(v - root_pos, root_q, transform)
            out.append(v)
        return np.concatenate(out)

    def get_body_quat(self):
        """Root quaternion + one quaternion per body from its hinge triple, quaternion_from_euler(z, y, x, 'rzyx') (humanoid_im.py:925-947)."""
        from ..utils.transformation import quaternion_from_euler
        qpos = self.get_humanoid_qpos()
        if self.vec.use_quat:  # :927-935: the ball joints' quaternions as they are
            return qpos[3:7 + 4 * (self.vec.body_lim - 2)].copy()
        out = [qpos[3:7]]
        for b in range(2, self.vec.body_lim):
            a = 7 + 3 * (b - 2)
            out.append(quaternion_from_euler(qpos[a], qpos[a + 1], qpos[a + 2], "rzyx"))
        return np.concatenate(out)

    def get_wbody_quat(self, selectList=None):
        d = self.data
        if selectList is None:
            return d.body_xquat[1:self.vec.body_lim].copy().ravel()
        return np.concatenate([d.body_xquat[self.model._body_name2id[b]] for b in selectList])

    def get_com(self):
        return self.data.get_body_xipos("Pelvis")

    def get_body_com(self, selectList=None):
        d = self.data
        if selectList is None:
            return d.xipos[1:self.vec.body_lim].copy().ravel()
        return np.concatenate([d.get_body_xipos(b) for b in selectList])

    def calc_body_diff(self):
        return float(self.vec.env.field(S.E_BODY_DIFF)[0].item())
