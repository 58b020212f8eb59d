"""Copycat (UHC) training configuration (mirror of uhc/utils/config_utils/copycat_config.py:12-168):
same attribute names and defaults, plus the knobs of the batched MI355X env (n_env, ppo_dtype)."""
import numpy as np

from .base_config import Base_Config


def _pad_edge(a, n):
    a = np.array(a)
    return np.pad(a, (0, max(0, n - a.size)), "edge")


class Config(Base_Config):
    def __init__(self, mujoco_path="%s.xml", **kwargs):
        super().__init__(**kwargs)
        c = self.cfg_dict
        g = c.get
        # training
        self.gamma, self.tau = g("gamma", 0.95), g("tau", 0.95)
        self.policy_htype, self.policy_hsize = g("policy_htype", "relu"), g("policy_hsize", [300, 200])
        self.policy_optimizer, self.policy_lr = g("policy_optimizer", "Adam"), g("policy_lr", 5e-5)
        self.policy_momentum, self.policy_weightdecay = g("policy_momentum", 0.0), g("policy_weightdecay", 0.0)
        self.value_htype, self.value_hsize = g("value_htype", "relu"), g("value_hsize", [300, 200])
        self.value_optimizer, self.value_lr = g("value_optimizer", "Adam"), g("value_lr", 3e-4)
        self.value_momentum, self.value_weightdecay = g("value_momentum", 0.0), g("value_weightdecay", 0.0)
        self.adv_clip, self.clip_epsilon = g("adv_clip", np.inf), g("clip_epsilon", 0.2)
        self.log_std, self.fix_std = g("log_std", -2.3), g("fix_std", False)
        self.num_optim_epoch = g("num_optim_epoch", 10)
        self.min_batch_size = g("min_batch_size", 50000)
        self.mini_batch_size = g("mini_batch_size", self.min_batch_size)
        self.save_n_epochs = g("save_n_epochs", 100)
        self.reward_id, self.reward_weights = g("reward_id", "quat"), g("reward_weights", None)
        self.end_reward = g("end_reward", False)
        self.actor_type = g("actor_type", "gauss")
        if self.actor_type == "mcp":
            self.num_primitive, self.composer_dim = g("num_primitive", 8), g("composer_dim", [[300, 200]])
        # adaptive schedules (piece-wise linear between check points)
        self.adp_iter_cp = np.array(g("adp_iter_cp", [0]))
        n = self.adp_iter_cp.size
        self.adp_noise_rate_cp = _pad_edge(g("adp_noise_rate_cp", [1.0]), n)
        self.adp_log_std_cp = _pad_edge(g("adp_log_std_cp", [self.log_std]), n)
        self.adp_policy_lr_cp = _pad_edge(g("adp_policy_lr_cp", [self.policy_lr]), n)
        self.adp_noise_rate = self.adp_log_std = self.adp_policy_lr = None
        # env
        self.mujoco_model = g("mujoco_model", "humanoid_smpl_neutral_mesh")
        self.mujoco_model_file = mujoco_path % self.mujoco_model
        self.vis_model_file = mujoco_path % g("vis_model", self.mujoco_model)
        self.env_start_first = g("env_start_first", False)
        self.env_init_noise = g("env_init_noise", 0.0)
        self.env_episode_len = g("env_episode_len", 200)
        self.env_term_body = g("env_term_body", "head")
        self.env_expert_trail_steps = g("env_expert_trail_steps", 0)
        self.obs_v, self.obs_type, self.obs_coord = g("obs_v", 0), g("obs_type", "full"), g("obs_coord", "root")
        self.obs_phase, self.obs_heading, self.obs_vel = g("obs_phase", True), g("obs_heading", False), g("obs_vel", "full")
        self.root_deheading = g("root_deheading", False)
        self.action_type, self.action_v = g("action_type", "position"), g("action_v", 0)
        self.reactive_v, self.reactive_rate, self.no_root = g("reactive_v", 0), g("reactive_rate", 0.3), g("no_root", False)
        self.sampling_temp, self.sampling_freq = g("sampling_temp", 0.2), g("sampling_freq", 0.75)
        # residual ("virtual") force
        self.residual_force = g("residual_force", False)
        self.residual_force_scale = g("residual_force_scale", 200.0)
        self.residual_force_lim = g("residual_force_lim", 100.0)
        self.residual_force_mode = g("residual_force_mode", "implicit")
        self.residual_force_bodies = g("residual_force_bodies", "all")
        self.residual_force_torque = g("residual_force_torque", True)
        self.rfc_decay = g("rfc_decay", False)
        # meta PD, misc
        self.meta_pd, self.meta_pd_joint = g("meta_pd", False), g("meta_pd_joint", False)
        self.masterfoot, self.fail_safe = g("masterfoot", False), g("fail_safe", True)
        self.robot_cfg = g("robot", {})
        if len(self.robot_cfg) == 0:
            self.robot_cfg = {"model": "smpl", "mesh": "mesh" in self.mujoco_model_file}
        self.has_shape = g("has_shape", False)
        # explicit gain / body-weight tables (copycat_config.py:128-145; parsed, though HumanoidEnv.load_models takes SMPLConverter's)
        if "joint_params" in c:
            jp = [np.array(p) for p in zip(*c["joint_params"])]
            self.jkp, self.jkd, self.a_ref, self.a_scale, self.torque_lim = jp[1:6]
            self.a_ref = np.deg2rad(self.a_ref)
            kpm = g("jkp_multiplier", 1.0)
            self.jkp = self.jkp * kpm
            self.jkd = self.jkd * g("jkd_multiplier", kpm)
            self.torque_lim = self.torque_lim * g("torque_limit_multiplier", 1.0)
        if "body_params" in c:
            self.b_diffw = [np.array(p) for p in zip(*c["body_params"])][1]
            self.jpos_diffw = np.concatenate([[1], self.b_diffw])
        self.agent_name, self.model_name = g("agent_name", "agent_copycat"), g("model_name", "super_net")
        # batched-env knobs of this build (not in the reference)
        # contact solve of the dual QP (UhcModelDesc.solver).  1 (default): its exact optimum by active-set iterations (block principal
        # pivoting, ~3 factorisations of the free block) -- what the reference's Newton solver converges to; envs beyond the fast
        # kernel's capacity are solved by sweeps.  0: projected Gauss-Seidel sweeps as MuJoCo's PGS, capped at pgs_iterations:
        # ~140 sweeps on average (up to ~400) reach the 1e-8 tolerance on this model, which still leaves the forces ~1e-4 (relative)
        # from the optimum; stopping at MuJoCo's default 100 leaves ~5e-3 of trajectory error over 200 control steps
        self.contact_solver = g("contact_solver", 1)
        self.pgs_iterations = g("pgs_iterations", 300)
        self.n_env = g("n_env", 1024)
        self.ppo_dtype = g("ppo_dtype", "float64")
        # data-parallel gradient exchange: the dtype on the wire.  float32 (default): SURVEY 8e's 32 MB per optimisation epoch; the local
        # gradient means travel scaled by the rank's sample count and are divided by the global count in the parameters' own float64, so the
        # N-rank update stays within 1e-4 relative of the single-process one over a whole 10-epoch update (tests/test_learner_cpu.py).
        # float64 = the learner's own arithmetic on the wire: the N-rank update equals the single-process one to 1e-10, at twice the bytes
        self.grad_allreduce_dtype = g("grad_allreduce_dtype", "float32")
        self.overlap_grad_exchange = bool(g("overlap_grad_exchange", True))  # value half of the exchange travels during the surrogate's backward pass

    def update_adaptive_params(self, i_iter):
        cp = self.adp_iter_cp
        ind = np.where(i_iter >= cp)[0][-1]
        nind = ind + int(ind < len(cp) - 1)
        t = (i_iter - cp[ind]) / (cp[nind] - cp[ind]) if nind > ind else 0.0
        self.adp_noise_rate = self.adp_noise_rate_cp[ind] * (1 - t) + self.adp_noise_rate_cp[nind] * t
        self.adp_log_std = self.adp_log_std_cp[ind] * (1 - t) + self.adp_log_std_cp[nind] * t
        self.adp_policy_lr = self.adp_policy_lr_cp[ind] * (1 - t) + self.adp_policy_lr_cp[nind] * t
