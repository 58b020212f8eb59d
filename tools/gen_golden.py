"""Generate tests/golden/*.npz by IMPORTING the reference (build container only).

Every fixture is pure data: seeded inputs + the outputs the reference's own functions produce.
The reference is pure Python; third-party packages it imports but this image lacks are stubbed
(tools/ref_import.py).  MuJoCo itself is never executed (it is absent): reference functions that
read `self.data` / `self.model` are called unbound on duck-typed objects carrying synthetic arrays.

    python tools/gen_golden.py            # writes all fixtures
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402

ref_import.install()
import uhc as _ref_uhc  # noqa: E402

ref_import.assert_is_reference(_ref_uhc)  # this repository also has a package called `uhc` (an alias of uhc_amd): never generate vectors from it
import torch  # noqa: E402

torch.set_default_dtype(torch.float64)  # scripts/train_uhc.py:80-81
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

from uhc_amd.model.mjcf import compile_mjcf_file  # noqa: E402

REF = ref_import.REF
MODEL = compile_mjcf_file(os.path.join(REF, "assets/mujoco_models/humanoid_smpl_neutral_mesh.xml"))


class DuckModel:
    """What the reference reads from a mujoco-py model, served from this build's compiled model."""

    def __init__(self, m):
        self.body_names = list(m.body_names)
        self.body_pos = m.body_pos.copy()
        self.body_ipos = m.body_ipos.copy()
        self.body_parentid = m.body_parentid.copy()
        self.body_jntadr = m.body_jntadr.copy()
        self.body_jntnum = m.body_jntnum.copy()
        self.jnt_qposadr = m.jnt_qposadr.copy()
        self.jnt_dofadr = m.jnt_dofadr.copy()
        self.nq, self.nv, self.nu = m.nq, m.nv, m.nu
        self._body_name2id = {n: i for i, n in enumerate(m.body_names)}
        self.actuator_ctrlrange = np.zeros((m.nu, 2))
        self.geom_bodyid = m.geom_bodyid.copy()
        self.opt = types.SimpleNamespace(timestep=m.timestep)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print("wrote", name, {k: np.asarray(v).shape for k, v in arrs.items()})


# --------------------------------------------------------------------------- G1 quaternion / heading helpers
def g1_math():
    from uhc.utils import math_utils as mu
    from uhc.utils import transformation as tf
    rng = np.random.default_rng(101)
    n = 64
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    p = rng.normal(size=(n, 4)); p /= np.linalg.norm(p, axis=1, keepdims=True)
    v = rng.normal(size=(n, 3))
    eul = rng.uniform(-np.pi, np.pi, size=(n, 3))
    out = dict(q=q, p=p, v=v, eul=eul)
    out["quaternion_multiply"] = np.array([tf.quaternion_multiply(a, b) for a, b in zip(q, p)])
    out["quaternion_inverse"] = np.array([tf.quaternion_inverse(a * 1.3) for a in q])
    out["quaternion_matrix"] = np.array([tf.quaternion_matrix(a)[:3, :3] for a in q])
    out["quaternion_from_euler_rzyx"] = np.array([tf.quaternion_from_euler(e[0], e[1], e[2], "rzyx") for e in eul])
    out["rotation_from_quaternion"] = np.array([tf.rotation_from_quaternion(a) for a in q])
    out["get_heading"] = np.array([mu.get_heading(a) for a in q])
    out["get_heading_q"] = np.array([mu.get_heading_q(a) for a in q])
    out["de_heading"] = np.array([mu.de_heading(a) for a in q])
    out["transform_vec_root"] = np.array([mu.transform_vec(b, a, "root") for a, b in zip(q, v)])
    out["transform_vec_heading"] = np.array([mu.transform_vec(b, a, "heading") for a, b in zip(q, v)])
    out["quat_mul_vec"] = np.array([mu.quat_mul_vec(a, b) for a, b in zip(q, v)])
    vb = rng.normal(size=(24, 3))
    out["vb"] = vb
    out["transform_vec_batch_root"] = mu.transform_vec_batch(vb, q[0], "root")  # shape (3, 24): see SURVEY 3.5
    nq1, nq0 = q[:24].ravel(), p[:24].ravel()
    out["multi_quat_diff"] = mu.multi_quat_diff(nq1, nq0)
    out["multi_quat_norm"] = mu.multi_quat_norm(out["multi_quat_diff"])
    out["get_angvel_fd"] = mu.get_angvel_fd(nq0, nq1, 1.0 / 30)
    # doctest known answers of transformation.py
    out["kat_about_axis"] = tf.quaternion_about_axis(0.123, [1, 0, 0])
    out["kat_from_euler"] = tf.quaternion_from_euler(1, 2, 3, "ryxz")
    out["kat_multiply"] = tf.quaternion_multiply([4, 1, -2, 3], [8, -5, 6, 7])
    save("g1_math", **out)


# --------------------------------------------------------------------------- G2/G3 AMASS pose -> qpos -> expert features
def make_clip(rng, T=40):
    """Synthetic clip: the shipped standing pose with smooth joint-space perturbations."""
    import joblib
    d = joblib.load(os.path.join(REF, "sample_data/standing_neutral.pkl"))
    base = d["pose_aa"][10].copy()
    t = np.arange(T)[:, None] / 30.0
    amp = rng.uniform(0, 0.3, size=(1, 72)); freq = rng.uniform(0.2, 2.0, size=(1, 72)); ph = rng.uniform(0, 2 * np.pi, size=(1, 72))
    pose = base[None] + amp * np.sin(2 * np.pi * freq * t + ph)
    pose[:, :3] = base[:3]  # keep the root orientation (axis-angle composition is not additive)
    trans = np.zeros((T, 3)); trans[:, 0] = 0.3 * np.sin(t[:, 0]); trans[:, 2] = 0.91437225 - MODEL.body_pos[1][2]
    return pose, trans


def g2_g3_expert():
    from uhc.smpllib.smpl_mujoco import smpl_to_qpose
    from uhc.smpllib.torch_smpl_humanoid import Humanoid
    import joblib
    dm = DuckModel(MODEL)
    d = joblib.load(os.path.join(REF, "sample_data/standing_neutral.pkl"))
    qpos_std = smpl_to_qpose(d["pose_aa"], dm, trans=None, count_offset=True)
    save("g2_smpl_to_qpose_standing", pose_aa=d["pose_aa"], qpos=qpos_std, ref_qpos=d["qpos"])
    rng = np.random.default_rng(202)
    pose, trans = make_clip(rng, 40)
    qpos = smpl_to_qpose(pose, dm, trans=trans, count_offset=True)
    h = Humanoid(model=dm)
    feat = h.qpos_fk(torch.from_numpy(qpos))
    feat = {k: np.asarray(v) for k, v in feat.items()}
    save("g3_qpos_fk", pose_aa=pose, trans=trans, **{("f_" + k): v for k, v in feat.items()})
    return dm, qpos, feat


# --------------------------------------------------------------------------- G4 observation / termination, G6 reward
class Cfg(dict):
    __getattr__ = dict.__getitem__

    def get(self, k, d=None):
        return dict.get(self, k, d)


def fake_env(dm, feat, rng, cur_t=5, shape=True):
    from uhc.envs.humanoid_im import HumanoidEnv
    from uhc.smpllib.smpl_mujoco import SMPLConverter
    env = object.__new__(HumanoidEnv)
    conv = SMPLConverter(dm, dm, smpl_model="smpl")
    env.model = dm
    env.converter = conv
    env.qpos_lim, env.qvel_lim, env.body_lim = conv.get_new_qpos_lim(), conv.get_new_qvel_lim(), conv.get_new_body_lim()
    env.jpos_diffw = conv.get_new_diff_weight()[:, None]
    env.body_diffw = conv.get_new_diff_weight()[1:]
    env.body_qposaddr = __import__("uhc.khrylib.utils.mujoco", fromlist=["x"]).get_body_qposaddr(dm)
    env.use_quat = False
    env.base_rot = [0.7071, 0.7071, 0.0, 0.0]
    env.no_root = False
    env.cur_t, env.start_ind = cur_t, 0
    env.frame_skip = 15  # dt is a property: model.opt.timestep * frame_skip (mujoco_env.py:115-117)
    env.ndof, env.vf_dim, env.meta_pd_dim = 69, 6, 30
    env.cc_cfg = Cfg(obs_coord="root", obs_vel="full", has_shape=shape, obs_v=2, residual_force=True,
                     env_expert_trail_steps=0, env_episode_len=100000,
                     reward_weights=dict(w_p=0.3, w_v=0.1, w_e=0.45, w_c=0.1, w_vf=0.05, k_p=2.0, k_v=0.005, k_e=5.0, k_c=100.0, k_vf=1.0))
    expert = dict(feat)
    expert["meta"] = {"cyclic": False}
    expert["beta"] = np.tile(rng.normal(size=(1, 16)), (feat["qpos"].shape[0], 1))
    expert["gender"] = np.full((feat["qpos"].shape[0],), 2.0)
    env.expert = expert
    # simulated state: expert frame cur_t perturbed
    T = feat["qpos"].shape[0]
    qpos = feat["qpos"][cur_t].copy()
    qpos[:3] += rng.normal(scale=0.05, size=3)
    dq = rng.normal(size=4) * 0.05
    qpos[3:7] = qpos[3:7] + dq; qpos[3:7] /= np.linalg.norm(qpos[3:7])
    qpos[7:] += rng.normal(scale=0.1, size=69)
    qvel = rng.normal(scale=0.5, size=75)
    xpos = np.vstack([np.zeros(3), feat["wbpos"][cur_t].reshape(-1, 3) + rng.normal(scale=0.03, size=(24, 3))])
    xq = feat["wbquat"][cur_t].reshape(-1, 4) + rng.normal(scale=0.05, size=(24, 4))
    xq /= np.linalg.norm(xq, axis=1, keepdims=True)
    xquat = np.vstack([np.array([1.0, 0, 0, 0]), xq])
    xipos = np.vstack([np.zeros(3), feat["body_com"][cur_t].reshape(-1, 3) + rng.normal(scale=0.03, size=(24, 3))])

    class Data:
        pass

    data = Data()
    data.qpos, data.qvel, data.body_xpos, data.body_xquat, data.xipos = qpos, qvel, xpos, xquat, xipos
    data.get_body_xipos = lambda name: xipos[dm._body_name2id[name]]
    env.data = data
    return env


def g4_g6_obs_reward(dm, feat):
    from uhc.envs.humanoid_im import HumanoidEnv
    from uhc.losses.reward_function import world_rfc_implicit_reward
    rng = np.random.default_rng(303)
    cases = {}
    for c, cur_t in enumerate([3, 7, 20, 38]):
        env = fake_env(dm, feat, rng, cur_t=cur_t)
        obs = HumanoidEnv.get_full_obs_v2(env)
        bquat = HumanoidEnv.get_body_quat(env)
        body_diff = HumanoidEnv.calc_body_diff(env)
        ee = HumanoidEnv.get_ee_pos(env, None)
        # reward needs prev_bquat (bquat before the step) and an action
        prev_q = env.data.qpos.copy()
        prev_q[7:] -= rng.normal(scale=0.02, size=69)
        save_q = env.data.qpos
        env.data.qpos = prev_q
        env.prev_bquat = HumanoidEnv.get_body_quat(env)
        env.data.qpos = save_q
        action = rng.normal(scale=0.3, size=105)
        r, rinfo = world_rfc_implicit_reward(env, None, action, None)
        pre = f"c{c}_"
        cases.update({pre + "cur_t": cur_t, pre + "qpos": env.data.qpos, pre + "qvel": env.data.qvel, pre + "xpos": env.data.body_xpos,
                      pre + "xquat": env.data.body_xquat, pre + "xipos": env.data.xipos, pre + "obs": obs, pre + "bquat": bquat,
                      pre + "body_diff": body_diff, pre + "ee": ee, pre + "prev_bquat": env.prev_bquat, pre + "action": action,
                      pre + "reward": r, pre + "reward_info": rinfo, pre + "beta": env.expert["beta"][0]})
    cases["beta"] = env.expert["beta"][0]
    cases["gender"] = env.expert["gender"][0]
    cases["ncase"] = 4
    save("g4_g6_obs_reward", **cases)
    g4b_obs_variants(dm, feat)


def g4b_obs_variants(dm, feat):
    """Observation v1 / v6 and the explicit-RFC reward on the same kind of synthetic env state (own rng stream)."""
    from uhc.envs.humanoid_im import HumanoidEnv
    from uhc.losses.reward_function import world_rfc_explicit_reward
    rng = np.random.default_rng(909)
    cases = {}
    for c, cur_t in enumerate([2, 11, 38]):
        env = fake_env(dm, feat, rng, cur_t=cur_t)
        env.cc_cfg.update(obs_v=1)
        obs1 = HumanoidEnv.get_full_obs_v1(env)
        obs6 = HumanoidEnv.get_full_obs_v6(env)
        env.cc_cfg.update(fut_frames=3, skip=4)
        obs3 = HumanoidEnv.get_full_obs_v3(env)
        prev_q = env.data.qpos.copy()
        prev_q[7:] -= rng.normal(scale=0.02, size=69)
        save_q = env.data.qpos
        env.data.qpos = prev_q
        env.prev_bquat = HumanoidEnv.get_body_quat(env)
        env.data.qpos = save_q
        env.vf_bodies = list(dm.body_names[1:])
        env.body_vf_dim, env.vf_dim = 9, 24 * 9
        action = rng.normal(scale=0.1, size=69 + 216 + 30)
        r, rinfo = world_rfc_explicit_reward(env, None, action, None)
        pre = f"c{c}_"
        cases.update({pre + "cur_t": cur_t, pre + "qpos": env.data.qpos, pre + "qvel": env.data.qvel, pre + "xpos": env.data.body_xpos,
                      pre + "xquat": env.data.body_xquat, pre + "xipos": env.data.xipos, pre + "obs_v1": obs1, pre + "obs_v6": obs6, pre + "obs_v3": obs3,
                      pre + "prev_bquat": env.prev_bquat, pre + "action": action, pre + "reward_explicit": r, pre + "reward_explicit_info": rinfo, pre + "beta": env.expert["beta"][0]})
    cases["beta"] = env.expert["beta"][0]
    cases["gender"] = env.expert["gender"][0]
    cases["ncase"] = 3
    save("g4b_obs_variants", **cases)


def g4c_more_variants(dm, feat):
    """Observation v0 / v5 (v4 returns a (full, local, global) tuple that the stock agent cannot consume: not built) and the remaining reward ids (implicit_quat, v1_mul, explicit_mul, v2, v3), own rng stream."""
    from uhc.envs.humanoid_im import HumanoidEnv
    from uhc.losses import reward_function as RF
    rng = np.random.default_rng(1313)
    w_v23 = dict(k_p=0.6, k_wp=0.3, k_v=0.004, k_j=60.0, k_c=80.0, k_vf=0.7, w_p=0.25, w_wp=0.2, w_v=0.05, w_j=0.3, w_c=0.15, w_vf=0.05,
                 jpos_diffw=[float(x) for x in np.round(np.linspace(0.5, 1.5, 24), 3)])
    cases = {"w_v23_keys": np.array([k for k in w_v23 if k != "jpos_diffw"]), "w_v23_vals": np.array([w_v23[k] for k in w_v23 if k != "jpos_diffw"]),
             "w_v23_jpos_diffw": np.array(w_v23["jpos_diffw"])}
    for c, cur_t in enumerate([1, 9, 37]):
        env = fake_env(dm, feat, rng, cur_t=cur_t)
        env.expert["len"] = feat["qpos"].shape[0]
        env.cc_cfg.update(obs_heading=True, root_deheading=True, obs_phase=True)
        obs0 = HumanoidEnv.get_full_obs(env)
        obs5 = HumanoidEnv.get_full_obs_v5(env)
        prev_q = env.data.qpos.copy()
        prev_q[7:] -= rng.normal(scale=0.02, size=69)
        save_q = env.data.qpos
        env.data.qpos = prev_q
        env.prev_bquat = HumanoidEnv.get_body_quat(env)
        env.data.qpos = save_q
        action = rng.normal(scale=0.3, size=105)
        pre = f"c{c}_"
        cases.update({pre + "cur_t": cur_t, pre + "qpos": env.data.qpos, pre + "qvel": env.data.qvel, pre + "xpos": env.data.body_xpos,
                      pre + "xquat": env.data.body_xquat, pre + "xipos": env.data.xipos, pre + "obs_v0": obs0, pre + "obs_v5": obs5,
                      pre + "prev_bquat": env.prev_bquat, pre + "action": action, pre + "beta": env.expert["beta"][0]})
        for name in ("world_rfc_implicit_quat", "world_rfc_implicit_v1_mul"):
            r, info = RF.reward_func[name](env, None, action, None)
            cases.update({pre + name: r, pre + name + "_info": info})
        base_w = env.cc_cfg["reward_weights"]
        env.cc_cfg["reward_weights"] = w_v23
        for name in ("world_rfc_implicit_v2", "world_rfc_implicit_v3"):
            r, info = RF.reward_func[name](env, None, action, None)
            cases.update({pre + name: r, pre + name + "_info": info})
        env.cc_cfg["reward_weights"] = base_w
        env.vf_bodies = list(dm.body_names[1:])
        env.body_vf_dim, env.vf_dim = 9, 24 * 9
        action_e = rng.normal(scale=0.1, size=69 + 216 + 30)
        r, info = RF.reward_func["world_rfc_explicit_mul"](env, None, action_e, None)
        cases.update({pre + "action_explicit": action_e, pre + "world_rfc_explicit_mul": r, pre + "world_rfc_explicit_mul_info": info})
    cases["gender"] = env.expert["gender"][0]
    cases["ncase"] = 3
    save("g4c_more_variants", **cases)


def g4d_obs_v4():
    """get_full_obs_v4 (uhc/envs/humanoid_im.py:769-861): the observation with its global and its per-body local part separated -- returns (obs_full, local_obs,
    global_obs).  The expert features are those of fixture G3 (read back from it: nothing else is rewritten)."""
    from uhc.envs.humanoid_im import HumanoidEnv
    dm = DuckModel(MODEL)
    g3 = np.load(os.path.join(OUT, "g3_qpos_fk.npz"))
    feat = {k[2:]: g3[k] for k in g3.files if k.startswith("f_")}
    rng = np.random.default_rng(1414)
    cases = {}
    for c, cur_t in enumerate([2, 11, 36]):
        env = fake_env(dm, feat, rng, cur_t=cur_t)
        env.expert["len"] = feat["qpos"].shape[0]
        env.cc_cfg.update(obs_v=4)
        full, local, glob = HumanoidEnv.get_full_obs_v4(env)
        pre = f"c{c}_"
        cases.update({pre + "cur_t": cur_t, pre + "qpos": env.data.qpos, pre + "qvel": env.data.qvel, pre + "xpos": env.data.body_xpos, pre + "xquat": env.data.body_xquat,
                      pre + "obs_full": full, pre + "local_obs": local, pre + "global_obs": glob, pre + "beta": env.expert["beta"][0]})
    cases["gender"] = env.expert["gender"][0]
    cases["ncase"] = 3
    save("g4d_obs_v4", **cases)


# --------------------------------------------------------------------------- G5 stable PD + implicit residual force
def g5_pd(dm, feat):
    from uhc.envs import humanoid_im
    from uhc.envs.humanoid_im import HumanoidEnv
    from uhc.smpllib.smpl_mujoco import SMPLConverter
    rng = np.random.default_rng(404)
    env = fake_env(dm, feat, rng, cur_t=4)
    conv = env.converter
    env.jkp, env.jkd = conv.get_new_jkp(), conv.get_new_jkd()
    env.torque_lim = conv.get_new_torque_limit()
    env.sim_iter = 15
    env.cc_cfg.update(action_v=1, meta_pd=True, meta_pd_joint=False, residual_force_scale=100.0, residual_force_lim=100.0)
    env.rfc_rate = 1
    from uhc_amd.model.mjcf import mass_matrix_np
    M = mass_matrix_np(MODEL, env.data.qpos)  # a genuine (tree-structured) joint-space inertia, handed over densely
    C = rng.normal(scale=5.0, size=75)
    env.data.qfrc_bias = C
    env.data.qM = M  # the patched mj_fullM below copies this dense matrix out
    env.data.qfrc_applied = np.zeros(75)

    def fake_fullM(model, out, qM):
        out[:] = np.asarray(qM).ravel()

    humanoid_im.mjf = types.SimpleNamespace(mj_fullM=fake_fullM)
    action = rng.normal(scale=0.5, size=105)
    out = dict(M=M, C=C, action=action, qpos=env.data.qpos, qvel=env.data.qvel, target_base=env.expert["qpos"][env.cur_t + 1][7:])
    for it in (0, 7, 14):
        out[f"torque_{it}"] = HumanoidEnv.compute_torque(env, action, i_iter=it)
    vf = action[69:75].copy()
    HumanoidEnv.rfc_implicit(env, vf)
    out["qfrc_applied"] = env.data.qfrc_applied.copy()
    out["jkp"], out["jkd"], out["torque_lim"] = env.jkp, env.jkd, env.torque_lim
    save("g5_pd_rfc", **out)


# --------------------------------------------------------------------------- G7 ZFilter, G8 GAE / PPO pieces
def g7_g8_learner():
    from uhc.khrylib.utils.zfilter import ZFilter
    from uhc.khrylib.rl.core.common import estimate_advantages
    from uhc.khrylib.rl.core.policy_gaussian import PolicyGaussian
    from uhc.khrylib.rl.core.critic import Value
    from uhc.khrylib.models.mlp import MLP
    rng = np.random.default_rng(505)
    zf = ZFilter((12,), clip=5)
    xs = rng.normal(loc=1.0, scale=3.0, size=(200, 12))
    ys = np.array([zf(x) for x in xs])
    save("g7_zfilter", xs=xs, ys=ys, mean=zf.rs.mean, std=zf.rs.std, n=zf.rs.n, y_noupdate=zf(xs[0], update=False))
    N = 500
    rewards = torch.from_numpy(rng.uniform(0, 1, size=(N, 1)))
    masks = torch.from_numpy((rng.uniform(size=(N, 1)) > 0.05).astype(np.float64))
    values = torch.from_numpy(rng.normal(size=(N, 1)))
    adv, ret = estimate_advantages(rewards, masks, values, 0.95, 0.95)
    save("g8_gae", rewards=rewards.numpy(), masks=masks.numpy(), values=values.numpy(), advantages=adv.numpy(), returns=ret.numpy())
    # small actor-critic (same classes, reduced width so the fixture stays small): forward, log-prob, PPO loss, grads
    torch.manual_seed(7)
    sd, ad = 23, 9
    pcfg = types.SimpleNamespace(policy_hsize=(32, 16), policy_htype="gelu", fix_std=True, log_std=-2.3)
    pol = PolicyGaussian(pcfg, action_dim=ad, state_dim=sd)  # policy_gaussian.py:9-24
    val = Value(MLP(sd, (32, 16), "gelu"))
    x = torch.from_numpy(rng.normal(size=(64, sd)))
    a = torch.from_numpy(rng.normal(scale=0.2, size=(64, ad)))
    advs = torch.from_numpy(rng.normal(size=(64, 1)))
    rets = torch.from_numpy(rng.normal(size=(64, 1)))
    with torch.no_grad():
        fixed_lp = pol.get_log_prob(x, a) + 0.01 * torch.from_numpy(rng.normal(size=(64, 1)))
    lp = pol.get_log_prob(x, a)
    ratio = torch.exp(lp - fixed_lp)
    surr1 = ratio * advs
    surr2 = torch.clamp(ratio, 0.8, 1.2) * advs
    loss = -torch.min(surr1, surr2).mean()  # agent_ppo.py:58-65
    loss.backward()
    vloss = (val(x) - rets).pow(2).mean()  # agent_pg.py:18-25
    vloss.backward()
    arrs = dict(x=x.numpy(), a=a.numpy(), advs=advs.numpy(), rets=rets.numpy(), fixed_lp=fixed_lp.numpy(), log_prob=lp.detach().numpy(),
                mean=pol(x).loc.detach().numpy(), value=val(x).detach().numpy(), ppo_loss=loss.item(), value_loss=vloss.item())
    for n, p in pol.named_parameters():
        arrs["pol_" + n] = p.detach().numpy()
        if p.grad is not None:
            arrs["polgrad_" + n] = p.grad.numpy()
    for n, p in val.named_parameters():
        arrs["val_" + n] = p.detach().numpy()
        arrs["valgrad_" + n] = p.grad.numpy()
    save("g8_ppo_small", **arrs)


def g8b_ppo_update():
    """The reference's own AgentPPO.update_policy (uhc/khrylib/rl/agents/agent_ppo.py:16-51) run end to end on a small actor-critic: three optimisation
    epochs of the full-batch branch and of the MINI-BATCH branch (`use_mini_batch`, :23-43: numpy-global-RNG permutation per epoch, applied cumulatively;
    floor(N / mini_batch_size) Adam steps of value and policy each) -- start parameters, inputs and the parameters after the update."""
    from uhc.khrylib.rl.agents.agent_ppo import AgentPPO
    from uhc.khrylib.rl.core.policy_gaussian import PolicyGaussian
    from uhc.khrylib.rl.core.critic import Value
    from uhc.khrylib.models.mlp import MLP
    rng = np.random.default_rng(808)
    sd, ad, N = 17, 6, 70
    x = torch.from_numpy(rng.normal(size=(N, sd)))
    a = torch.from_numpy(rng.normal(scale=0.2, size=(N, ad)))
    advs = torch.from_numpy(rng.normal(size=(N, 1)))
    rets = torch.from_numpy(rng.normal(size=(N, 1)))
    exps = torch.from_numpy((rng.uniform(size=N) > 0.2).astype(np.float64))
    arrs = dict(x=x.numpy(), a=a.numpy(), advs=advs.numpy(), rets=rets.numpy(), exps=exps.numpy(), mini_batch_size=16, epochs=3, np_seed=33,
                policy_lr=5e-3, value_lr=3e-3, clip_epsilon=0.2, grad_clip=0.5)
    for tag, mini in (("full", False), ("mini", True)):
        torch.manual_seed(21)
        pcfg = types.SimpleNamespace(policy_hsize=(24, 12), policy_htype="gelu", fix_std=True, log_std=-2.3)
        pol = PolicyGaussian(pcfg, action_dim=ad, state_dim=sd)
        val = Value(MLP(sd, (24, 12), "gelu"))
        if not mini:
            for n, p in pol.named_parameters():
                arrs["pol0_" + n] = p.detach().numpy().copy()
            for n, p in val.named_parameters():
                arrs["val0_" + n] = p.detach().numpy().copy()
        opt_p = torch.optim.Adam([p for p in pol.parameters() if p.requires_grad], lr=5e-3)
        opt_v = torch.optim.Adam(val.parameters(), lr=3e-3)
        ag = AgentPPO(env=None, policy_net=pol, value_net=val, dtype=torch.float64, device=torch.device("cpu"), gamma=0.95, data_loader=None,
                      optimizer_policy=opt_p, optimizer_value=opt_v, opt_num_epochs=3, value_opt_niter=1, clip_epsilon=0.2, mini_batch_size=16, use_mini_batch=mini,
                      policy_grad_clip=[(list(pol.parameters()), 0.5)])  # (a LIST here: the generator quirk of agent_copycat.py is pinned elsewhere)
        np.random.seed(33)
        ag.update_policy(x.clone(), a.clone(), rets.clone(), advs.clone(), exps.clone())
        for n, p in pol.named_parameters():
            arrs[f"pol_{tag}_" + n] = p.detach().numpy().copy()
        for n, p in val.named_parameters():
            arrs[f"val_{tag}_" + n] = p.detach().numpy().copy()
    save("g8b_ppo_update", **arrs)


def g12_policy_mcp():
    """PolicyMCP (uhc/models/policy_mcp.py:9-37) at reduced width: parameters, mean, log-prob and gradients."""
    from uhc.models.policy_mcp import PolicyMCP
    rng = np.random.default_rng(1212)
    torch.manual_seed(12)
    sd, ad = 19, 7

    class C(dict):
        __getattr__ = dict.__getitem__

    cfg = C(policy_hsize=[24, 16, 12], policy_htype="gelu", fix_std=True, log_std=-2.3, num_primitive=4, composer_dim=[20, 10])
    pol = PolicyMCP(cfg, action_dim=ad, state_dim=sd)
    for n, p in pol.named_parameters():  # heads are near zero at init; make every block matter
        if p.requires_grad:
            p.data.add_(torch.from_numpy(rng.normal(scale=0.05, size=tuple(p.shape))))
    x = torch.from_numpy(rng.normal(size=(32, sd)))
    a = torch.from_numpy(rng.normal(scale=0.2, size=(32, ad)))
    lp = pol.get_log_prob(x, a)
    (-lp.mean()).backward()
    arrs = dict(x=x.numpy(), a=a.numpy(), mean=pol(x).loc.detach().numpy(), log_prob=lp.detach().numpy(), weight=pol.composer(x).detach().numpy())
    for n, p in pol.named_parameters():
        arrs["pol_" + n] = p.detach().numpy()
        if p.grad is not None:
            arrs["polgrad_" + n] = p.grad.numpy()
    save("g12_policy_mcp", **arrs)


def g11_metrics(dm, feat):
    from uhc.smpllib.smpl_eval import compute_metrics
    rng = np.random.default_rng(606)
    T = 30
    gt, gj = feat["qpos"][:T].copy(), feat["wbpos"][:T].copy()
    pred = gt + rng.normal(scale=0.02, size=gt.shape)
    pred[:, 3:7] /= np.linalg.norm(pred[:, 3:7], axis=1, keepdims=True)
    pj = gj + rng.normal(scale=0.03, size=gj.shape)
    res = dict(pred=pred, gt=gt, pred_jpos=pj, gt_jpos=gj, fail_safe=False, percent=1)
    out = compute_metrics(res, None)
    save("g11_metrics", pred=pred, gt=gt, pred_jpos=pj, gt_jpos=gj, **{("m_" + k): np.asarray(v) for k, v in out.items()})


# --------------------------------------------------------------------------- G9 clip sampling
def synth_pickle(rng):
    """8 small clips: mixed lengths (one below t_min+1), beta 10/16 wide, three genders."""
    pk = {}
    lens = [70, 25, 41, 16, 55, 33, 90, 12]
    for i, T in enumerate(lens):
        pk[f"0-SYN_{i:02d}_poses"] = dict(pose_aa=rng.normal(size=(T, 72)) * 0.1, pose_6d=np.zeros((T, 144)), trans=rng.normal(size=(T, 3)),
                                          beta=rng.normal(size=(T, 10)) if i % 2 else rng.normal(size=(16,)), gender=["neutral", "male", "female"][i % 3], seq_name=f"SYN_{i:02d}")
    return pk


def g9_dataset():
    import random
    import tempfile
    import joblib
    from uhc.data_loaders.dataset_amass_single import DatasetAMASSSingle
    rng = np.random.default_rng(9)
    pk = synth_pickle(rng)
    d = tempfile.mkdtemp()
    joblib.dump(pk, os.path.join(d, "syn.pkl"))
    specs = dict(file_path=os.path.join(d, "syn.pkl"), test_file_path=os.path.join(d, "syn.pkl"), t_min=15, t_max=30, mode="all",
                 neutral_path=os.path.join(REF, "sample_data/standing_neutral.pkl"))
    dl = DatasetAMASSSingle(specs, "train")
    keys = list(pk.keys())
    out = {"n_data_keys": len(dl.data_keys), "data_keys": np.array([keys.index(k) for k in dl.data_keys]),
           "sample_keys": np.array([keys.index(k) for k, _ in dl.sample_keys])}
    for i, k in enumerate(keys):
        for f in ("pose_aa", "pose_6d", "trans", "beta"):
            out[f"in{i}_{f}"] = pk[k][f]
    # (a) uniform mode
    random.seed(3); np.random.seed(3)
    seq = []
    for _ in range(24):
        s = dl.sample_seq(freq_dict=None)
        seq.append([keys.index(dl.curr_key), dl.fr_start, dl.fr_end, s["pose_aa"].shape[0], s["beta"].shape[1], s["gender"][0], int(s["has_obj"]), s["num_obj"]])
    out["uniform"] = np.array(seq)
    out["uniform_first_pose"] = s["pose_aa"]
    # (b) success-weighted mode with a filled freq_dict, (c) precision mode
    frng = np.random.default_rng(4)
    fd = {k: [[float(frng.random() < 0.6) if frng.random() < 0.8 else float(frng.random()), int(frng.integers(0, 10))] for _ in range(int(frng.integers(0, 9)))] for k in dl.data_keys}
    out["fd_flat"] = np.array([[dl.data_keys.index(k), p, s0] for k, v in fd.items() for p, s0 in v])
    for name, prec in (("weighted", False), ("precision", True)):
        random.seed(5); np.random.seed(5)
        seq = []
        for _ in range(24):
            dl.sample_seq(freq_dict=fd, sampling_temp=0.2, sampling_freq=0.5, precision_mode=prec)
            seq.append([keys.index(dl.curr_key), dl.fr_start, dl.fr_end])
        out[name] = np.array(seq)
    s = dl.get_sample_from_key(dl.data_keys[2], full_sample=True)
    out["full"] = np.array([dl.fr_start, dl.fr_end, s["pose_aa"].shape[0]])
    s = dl.get_sample_from_key(dl.data_keys[0], fr_start=7)
    out["fixed_start"] = np.array([dl.fr_start, dl.fr_end, s["pose_aa"].shape[0]])
    out["iter"] = np.array([keys.index(dl.iter_seq()["seq_name"]) for _ in range(8)])
    save("g9_dataset", **out)


# --------------------------------------------------------------------------- G10 configs
def g10_config():
    import json
    import tempfile
    import yaml
    import glob
    from uhc.utils.config_utils.copycat_config import Config
    ids = ["uhc_implicit", "uhc_implicit_shape", "uhc_explicit", "copycat_ball_1", "copycat_40", "copycat_44"]
    scalars = ["gamma", "tau", "policy_htype", "policy_hsize", "policy_lr", "value_htype", "value_hsize", "value_lr", "clip_epsilon", "log_std", "fix_std",
               "num_optim_epoch", "min_batch_size", "mini_batch_size", "save_n_epochs", "reward_id", "reward_weights", "end_reward", "actor_type",
               "env_start_first", "env_init_noise", "env_episode_len", "env_term_body", "env_expert_trail_steps", "obs_v", "obs_type", "obs_coord",
               "obs_phase", "obs_heading", "obs_vel", "root_deheading", "action_type", "action_v", "reactive_v", "no_root", "reactive_rate",
               "sampling_temp", "sampling_freq", "residual_force", "residual_force_scale", "residual_force_lim", "residual_force_mode",
               "residual_force_bodies", "residual_force_torque", "rfc_decay", "meta_pd", "meta_pd_joint", "masterfoot", "fail_safe", "robot_cfg",
               "has_shape", "agent_name", "model_name", "seed", "notes", "num_epoch", "proj_name", "num_primitive", "composer_dim", "data_specs", "lr",
               "eval_n_epochs", "num_samples", "batch_size"]
    arrays = ["adp_iter_cp", "adp_noise_rate_cp", "adp_log_std_cp", "adp_policy_lr_cp", "jkp", "jkd", "a_ref", "a_scale", "torque_lim", "b_diffw", "jpos_diffw"]
    out = {}
    for cid in ids + ["synthetic_adaptive"]:
        if cid == "synthetic_adaptive":  # none of the shipped configs has a multi-point schedule; exercise update_adaptive_params
            f, = glob.glob(os.path.join(REF, "config/**/uhc_implicit_shape.yml"), recursive=True)
            cd = yaml.safe_load(open(f))
            cd.update(adp_iter_cp=[0, 100, 1000, 5000], adp_noise_rate_cp=[1.0, 0.5], adp_log_std_cp=[-2.3, -3.0, -3.5], adp_policy_lr_cp=[5e-5, 1e-5])
        else:
            f, = glob.glob(os.path.join(REF, f"config/**/{cid}.yml"), recursive=True)
            cd = yaml.safe_load(open(f))
        out[f"{cid}__yml"] = np.array(json.dumps(cd))
        base = tempfile.mkdtemp()  # results/ dirs are created under base_dir; assets are found through a link
        os.symlink(os.path.join(REF, "assets"), os.path.join(base, "assets"))
        cfg = Config(cfg_id=cid, base_dir=base, cfg_dict=json.loads(json.dumps(cd)))
        dump = {k: getattr(cfg, k) for k in scalars if hasattr(cfg, k)}
        dump["adv_clip_is_inf"] = bool(np.isinf(cfg.adv_clip))
        out[f"{cid}__scalars"] = np.array(json.dumps(dump))
        for k in arrays:
            if hasattr(cfg, k):
                out[f"{cid}__{k}"] = np.asarray(getattr(cfg, k), dtype=np.float64)
        ad = []
        for it in (0, 1, 50, 499, 1000, 2500, 5000, 20000):
            cfg.update_adaptive_params(it)
            ad.append([it, cfg.adp_noise_rate, cfg.adp_log_std, cfg.adp_policy_lr])
        out[f"{cid}__adaptive"] = np.array(ad, dtype=np.float64)
    save("g10_config", **out)


# --------------------------------------------------------------------------- G13 AMASS database processing
def g13_process_amass():
    """process_qpos_list + the split rule of uhc/data_process/process_amass_db.py on a synthetic AMASS db; the SMPL height fix
    (needs the licensed model files) is replaced by the identity on both sides."""
    import importlib
    mod = importlib.import_module("uhc.data_process.process_amass_db")
    rng = np.random.default_rng(1717)
    names = ["CMU_01_01_poses", "KIT_3_walk_poses", "SSM_synced_x_poses", "HumanEva_S1_poses", "ACCAD_sit_poses", "BMLmovi_air_poses",
             "DanceDB_short_poses", "SFU_bad_poses", "Unknown_set_poses", "MPI_mosh_tiny_poses"]
    db = {}
    for i, n in enumerate(names):
        T = [130, 240, 95, 60, 200, 150, 40, 120, 70, 25][i]
        fr = [120.0, 100.0, 60.0, 120.0, 60.0, 59.94, 30.0, 120.0, 60.0, 120.0][i]
        poses = rng.normal(scale=0.4, size=(T, 156))
        poses[3, 3:6] = 1e-5  # one joint in the small-angle branch
        db[n] = {"poses": poses, "trans": rng.normal(size=(T, 3)), "betas": rng.normal(size=16), "gender": ["male", "female", "neutral"][i % 3],
                 "mocap_framerate": fr}
        db[n]["poses"] = db[n]["poses"][:, :72]
    occ = {"0-ACCAD_sit_poses": {"issue": "sitting", "idxes": [35, 36]}, "0-BMLmovi_air_poses": {"issue": "airborne", "idxes": [6]},
           "0-SFU_bad_poses": {"issue": "stairs", "idxes": [3]}, "0-KIT_3_walk_poses": {"issue": "sitting"}}
    mod.target_fr = 30
    mod.amass_occlusion = occ
    mod.fix_height_smpl_vanilla = lambda pose_aa, th_betas, th_trans, gender, seq_name: th_trans
    mod.flags.debug = False
    res = mod.process_qpos_list(list(db.items()))
    out = {"names": np.array(names), "occ_keys": np.array(list(occ)), "occ_issue": np.array([occ[k]["issue"] for k in occ]),
           "occ_idx0": np.array([occ[k].get("idxes", [-1])[0] for k in occ]), "kept": np.array(list(res))}
    for n in names:
        for f in ("poses", "trans", "betas"):
            out[f"db_{n}_{f}"] = db[n][f]
        out[f"db_{n}_gender"] = np.array(db[n]["gender"])
        out[f"db_{n}_fr"] = db[n]["mocap_framerate"]
    for k, v in res.items():
        for f in ("pose_aa", "pose_6d", "trans", "beta"):
            out[f"res_{k}_{f}"] = np.asarray(v[f])
    # the split rule (the __main__ block, :340-361)
    split = {}
    for k in res:
        start_name = k.split("-")[1]
        for dataset_key in mod.amass_split_dict.keys():
            if start_name.lower().startswith(dataset_key.lower()):
                sp = mod.amass_split_dict[dataset_key]
                split[k] = "test" if sp == "test" else ("valid" if sp == "valid" else "train")
    out["split_keys"] = np.array(list(split))
    out["split_vals"] = np.array(list(split.values()))
    save("g13_process_amass", **out)


# --------------------------------------------------------------------------- G14 ball-joint env (robot.ball / use_quat)
def g14_ball_env():
    """The quaternion paths of the ball-joint humanoid (config/copycat_ball): `smpl_to_qpose(use_quat=True)` (:590-600), `get_body_quat`
    with use_quat (humanoid_im.py:927-935), `world_rfc_implicit_quat` on them, and `get_full_obs_v2_quat` (:668-756) with the
    QUATERNION expert pose behind `get_expert_qpos` -- the array `load_expert` computes as `expert_qpos_quat` (:193-200) but never
    stores: as shipped the function receives the 76-wide Euler pose and raises on `reshape(-1, 4)`.  The fixture therefore pins what
    the function computes when it is given the pose it was written for."""
    from uhc.envs.humanoid_im import HumanoidEnv
    from uhc.losses import reward_function as RF
    from uhc.smpllib.smpl_mujoco import smpl_to_qpose
    from uhc.smpllib.torch_smpl_humanoid import Humanoid
    from uhc_amd.model.mjcf import ball_variant
    ball = ball_variant(MODEL)
    dmb, dmh = DuckModel(ball), DuckModel(MODEL)
    rng = np.random.default_rng(1414)
    pose, trans = make_clip(rng, 40)
    qpos_e = smpl_to_qpose(pose, dmh, trans=trans, count_offset=True)                   # Euler expert (what qpos_fk consumes)
    qpos_q = smpl_to_qpose(pose, dmb, trans=trans, count_offset=True, use_quat=True)    # quaternion expert, (T, 99)
    feat = {k: np.asarray(v) for k, v in Humanoid(model=dmh).qpos_fk(torch.from_numpy(qpos_e)).items()}
    cases = {"pose_aa": pose, "trans": trans, "qpos_quat": qpos_q, "qpos_euler": qpos_e}
    for c, cur_t in enumerate([2, 9, 30]):
        env = fake_env(dmb, feat, rng, cur_t=cur_t)
        env.use_quat = True
        env.qpos_lim, env.qvel_lim, env.body_lim = 99, 75, 25
        env.expert = dict(env.expert)
        env.expert["qpos"] = qpos_q  # get_expert_qpos -> the quaternion pose
        env.cc_cfg.update(residual_force=False)
        env.vf_dim, env.meta_pd_dim = 0, 0
        q = qpos_q[cur_t].copy()
        q[:3] += rng.normal(scale=0.05, size=3)
        qq = q[3:].reshape(24, 4) + rng.normal(scale=0.05, size=(24, 4))
        q[3:] = (qq / np.linalg.norm(qq, axis=1, keepdims=True)).ravel()
        env.data.qpos = q
        obs = HumanoidEnv.get_full_obs_v2_quat(env)
        bquat = HumanoidEnv.get_body_quat(env)
        prev = q.copy()
        pq = prev[3:].reshape(24, 4) + rng.normal(scale=0.01, size=(24, 4))
        prev[3:] = (pq / np.linalg.norm(pq, axis=1, keepdims=True)).ravel()
        env.data.qpos = prev
        env.prev_bquat = HumanoidEnv.get_body_quat(env)
        env.data.qpos = q
        action = rng.normal(scale=0.3, size=69)
        r, info = RF.reward_func["world_rfc_implicit_quat"](env, None, action, None)
        pre = f"c{c}_"
        cases.update({pre + "cur_t": cur_t, pre + "qpos": q, pre + "qvel": env.data.qvel, pre + "xpos": env.data.body_xpos, pre + "xquat": env.data.body_xquat,
                      pre + "xipos": env.data.xipos, pre + "obs": obs, pre + "bquat": bquat, pre + "prev_bquat": env.prev_bquat, pre + "action": action,
                      pre + "reward": r, pre + "reward_info": info, pre + "beta": env.expert["beta"][0]})
    cases["gender"] = env.expert["gender"][0]
    cases["ncase"] = 3
    for k, v in feat.items():
        cases["f_" + k] = v
    save("g14_ball_env", **cases)


def main():
    if len(sys.argv) > 1:  # python tools/gen_golden.py g8b_ppo_update ...: only the named fixtures (the others are not rewritten)
        for name in sys.argv[1:]:
            globals()[name]()
        return
    g8b_ppo_update()
    g14_ball_env()
    g4d_obs_v4()  # (after g2_g3_expert in a full run would be tidier; it reads the committed G3 fixture, which a full run rewrites identically first below)
    g1_math()
    dm, qpos, feat = g2_g3_expert()
    g4_g6_obs_reward(dm, feat)
    g4c_more_variants(dm, feat)
    g5_pd(dm, feat)
    g7_g8_learner()
    g11_metrics(dm, feat)
    g9_dataset()
    g10_config()
    g12_policy_mcp()
    g13_process_amass()


if __name__ == "__main__":
    main()
