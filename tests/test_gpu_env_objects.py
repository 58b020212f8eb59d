"""Objects in the env layer (BASELINE configs[4]: "copycat_ball config with object contacts" as ONE rollout through env + policy).

The reference: `load_expert` / `reset_robot` hand expert["obj_info"] to the generator, which appends one free body + mesh geom per
object behind the humanoid (uhc/envs/humanoid_im.py:154-175, uhc/smpllib/smpl_robot.py:1200-1252); `reset_model` starts them at
expert["obj_pose"][0] with zero velocity (:1284-1287); observation, reward, termination and the PD controller read the humanoid in front
of them (qpos[:qpos_lim], qvel[:qvel_lim], M[:qvel_lim, :qvel_lim]: :421-422, :1021-1022); `get_obj_qpos` / `get_obj_qvel` are the rest
(:1423-1428).  GPU env kernels + fused physics through the C-ABI vs env_oracle.py + physics_oracle.c on the model WITH the objects."""
import dataclasses
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REWARD_W = dict(w_p=0.3, w_v=0.1, w_e=0.45, w_c=0.1, w_vf=0.05, k_p=2.0, k_v=0.005, k_e=5.0, k_c=100.0, k_vf=1.0)


def _hulls():
    from uhc_amd.model.shapes import box_triangles
    return [box_triangles(0.15, 0.15, 0.15), box_triangles(0.1, 0.2, 0.08)]


def _obj_rows(T, root_xy):
    """obj_pose (T, 14): two boxes near the humanoid, moving a little from frame to frame (the reset must read the WINDOW's first row)."""
    t = np.arange(T)[:, None]
    a = np.c_[root_xy[0] + 0.42 + 0.001 * t, root_xy[1] + 0.0 * t, 0.151 + 0.0 * t, np.ones((T, 1)), np.zeros((T, 3))]
    # (a generic orientation: a box tilted about ONE axis lands on an edge whose two vertices tie to the last bit, and which of them is "within
    #  the margin" in the first touching substep is then decided by rounding -- differently in two correct implementations)
    b = np.c_[root_xy[0] - 0.1 + 0.0 * t, root_xy[1] + 0.5 - 0.001 * t, 0.55 + 0.002 * t, np.full((T, 1), 0.9), np.full((T, 1), 0.1), np.full((T, 1), 0.07), np.full((T, 1), -0.04)]
    return np.concatenate([a, b], 1)


@pytest.mark.parametrize("ball", [False, True], ids=["hinge_pd", "ball_torque"])
def test_env_with_objects_matches_oracles(model, ctrl, ball):
    import torch
    from oracle import env_oracle as E
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd._capi import env_desc
    from uhc_amd.model.mjcf import add_free_bodies, ball_variant, self_collision_variant, trailing_free_bodies
    from uhc_amd.smpllib.smpl_mujoco import SMPLConverter
    from uhc_amd.smpllib.torch_smpl_humanoid import Humanoid
    rng = np.random.default_rng(23)
    beta = rng.normal(size=16)
    if ball:
        g = np.load(os.path.join(G, "g14_ball_env.npz"))
        body = self_collision_variant(ball_variant(model))
        c = S.make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
        euler, quatp = g["qpos_euler"], g["qpos_quat"]
        whole = Humanoid(model=model).qpos_fk(torch.from_numpy(euler.copy()))
        rec = np.array(whole["qpos"], copy=True)
        rec[:, 3:7] = quatp[:, 3:7]
        frames = S.pack_expert_frames(dict(whole, qpos=rec))
    else:
        f = np.load(os.path.join(G, "g3_qpos_fk.npz"))
        expert = {k[2:]: f[k] for k in f.files if k.startswith("f_")}
        expert["len"] = int(expert["len"])
        body = self_collision_variant(model)
        c = ctrl
        euler = expert["qpos"]
        frames = S.pack_expert_frames(expert)
    body = dataclasses.replace(body, solver=1)
    T = frames.shape[0]
    objs = _obj_rows(T, euler[0, :2])
    park = np.stack([np.r_[2.0 + k, 2.0, 1.0, 1, 0, 0, 0] for k in range(2)])
    full = add_free_bodies(body, _hulls(), park, density=400.0, friction=1.0, condim=3)
    assert trailing_free_bodies(full) == 2 and trailing_free_bodies(body) == 0
    nqh, nvh, nbh = int(body.nq), int(body.nv), int(body.nbody)
    n = 3
    sb = S.SimBatch(full, c, n)
    jw = SMPLConverter(model, model).get_new_diff_weight()
    eb = S.EnvBatch(sb, env_desc(model, obs_v=2, has_shape=True, reward_weights=REWARD_W, reward_v=0, jpos_diffw=jw, num_obj=2))
    assert eb.obs_dim == (534 if ball else 657)  # the objects are not part of the observation
    eb.set_bank(torch.from_numpy(frames), torch.tensor([0], dtype=torch.int32), torch.from_numpy(np.r_[beta, 2.0][None]))
    with pytest.raises(Exception, match="obj_pose"):  # a model with objects cannot be reset without their poses
        eb.reset(torch.arange(n, dtype=torch.int32).cuda(), None)
    eb.set_obj_pose(torch.from_numpy(objs))
    starts, lens = np.array([0, 5, 11]), np.array([28, 18, 9])
    ids = torch.arange(n, dtype=torch.int32)
    eb.assign(ids, torch.zeros(n, dtype=torch.int32), torch.from_numpy(starts), torch.from_numpy(lens))
    noise = None if ball else rng.normal(scale=0.03, size=(n, 69))
    eb.reset(ids.cuda(), None if ball else torch.from_numpy(noise))
    sb.sync()

    def window(s, l):
        w = Humanoid(model=model).qpos_fk(torch.from_numpy(euler[s:s + l].copy()))
        if ball:
            w["qpos_quat"] = quatp[s:s + l]
        return w

    wins = [window(starts[e], lens[e]) for e in range(n)]
    os_ = []
    gq0, gv0 = sb.field(S.F_QPOS).cpu().numpy(), sb.field(S.F_QVEL).cpu().numpy()
    for e in range(n):
        o = OracleSim(full, c)
        qh = wins[e]["qpos_quat"][0].copy() if ball else wins[e]["qpos"][0].copy()
        if not ball:
            qh[7:] += noise[e]
        q0 = np.r_[qh, objs[starts[e]]]           # init_pose = concat(expert pose, obj_pose[ind]) -- the window's first frame
        v0 = np.r_[wins[e]["qvel"][0], np.zeros(12)]  # init_vel = concat(expert velocity, zeros(6 num_obj))
        np.testing.assert_allclose(gq0[e], q0, atol=1e-15)
        np.testing.assert_allclose(gv0[e], v0, atol=1e-15)
        o.set_state(q0, v0)
        os_.append(o)

    def obs_of(o, w, t):
        xpos, xquat = o.get("xpos").reshape(-1, 3)[:nbh], o.get("xquat").reshape(-1, 4)[:nbh]
        fn = E.full_obs_v2_quat if ball else E.full_obs_v2
        return fn(o.get("qpos")[:nqh], o.get("qvel")[:nvh], xpos, xquat, w, t, 0, beta, 2.0)

    gobs = eb.field(S.E_OBS).cpu().numpy()
    for e in range(n):
        np.testing.assert_allclose(gobs[e], obs_of(os_[e], wins[e], 0), atol=1e-11)
    cur_t, alive = np.zeros(n, dtype=int), np.ones(n, dtype=bool)
    touched = 0
    for t in range(12):
        act = rng.normal(scale=0.003 if ball else 0.1, size=(n, c.action_dim))
        eb.step(torch.from_numpy(act).cuda(), torch.from_numpy(alive.astype(np.int32)).cuda())
        sb.sync()
        gobs, grew, gparts = eb.field(S.E_OBS).cpu().numpy(), eb.field(S.E_REWARD).cpu().numpy(), eb.field(S.E_REWARD_PARTS).cpu().numpy()
        gdone, gq, gv, redo = eb.field(S.E_DONE).cpu().numpy(), sb.field(S.F_QPOS).cpu().numpy(), sb.field(S.F_QVEL).cpu().numpy(), sb.field(S.F_REDO).cpu().numpy()
        for e in range(n):
            if not alive[e]:
                continue
            o, w = os_[e], wins[e]
            prev_bquat = E.get_body_quat_ball(o.get("qpos")[:nqh]) if ball else E.get_body_quat(o.get("qpos")[:nqh])
            tb = np.zeros(69) if ball else w["qpos"][E.expert_index(cur_t[e] + 1, 0, w["len"])][7:]
            o.do_simulation(act[e], tb, redo=redo[e])
            cur_t[e] += 1
            dq = np.abs(gq[e] - o.get("qpos"))
            assert dq.max() < 1e-9, (t, e, np.nonzero(dq > 1e-9)[0].tolist(), gq[e][dq > 1e-9].tolist(), o.get("qpos")[dq > 1e-9].tolist(), int(redo[e]), o.geti("ncon"), o.geti("nefc"))  # humanoid AND objects
            np.testing.assert_allclose(gv[e], o.get("qvel"), atol=1e-7)
            xpos, xipos = o.get("xpos").reshape(-1, 3)[:nbh], o.get("xipos").reshape(-1, 3)[:nbh]
            r, parts = E.world_rfc_implicit_reward(o.get("qpos")[:nqh], xpos, xipos, prev_bquat, act[e], w, cur_t[e], 0, model.timestep * 15, jw[1:], REWARD_W,
                                                   **(dict(vf_dim=0, ball=True) if ball else {}))
            bd = E.calc_body_diff(xpos, w["wbpos"][E.expert_index(cur_t[e], 0, w["len"])], jw)
            fail, end = bool(o.geti("fail")) or bd > 0.5, cur_t[e] >= w["len"] - 1
            assert bool(gdone[e]) == (fail or end)
            assert grew[e] == pytest.approx(r, abs=1e-9)
            np.testing.assert_allclose(gparts[e][:5], parts, atol=1e-9)
            np.testing.assert_allclose(gobs[e], obs_of(o, w, cur_t[e]), atol=1e-8)
            touched = max(touched, o.geti("ncon"))
            if fail or end:
                alive[e] = False
    assert not alive[2] and touched > 0
    # the objects moved (one was dropped from half a metre) and are where the oracle has them; get_obj_qpos = qpos[qpos_lim:]
    assert np.abs(gq[0][nqh:] - objs[starts[0]]).max() > 1e-3
    # auto_reset: a finished env restarts with the objects at its (queued) window's first row again
    eb.set_next(torch.tensor([2], dtype=torch.int32), torch.tensor([0], dtype=torch.int32), torch.tensor([7], dtype=torch.int32), torch.tensor([10], dtype=torch.int32), None)
    eb.auto_reset()
    sb.sync()
    q2, v2 = sb.field(S.F_QPOS).cpu().numpy()[2], sb.field(S.F_QVEL).cpu().numpy()[2]
    np.testing.assert_allclose(q2[nqh:], objs[7], atol=1e-15)
    assert (v2[nvh:] == 0).all()
    sb.close()


def test_agent_iteration_and_eval_on_the_ball_humanoid_with_objects(tmp_path):
    """configs[4] as the reference names it: copycat_ball's env (ball joints, torque actions, quaternion observation and reward) WITH
    objects, as one rollout through env + policy + PPO update + evaluation (mean-action episodes with the fail-safe teleport, which keeps
    the objects where they are: uhc/envs/humanoid_im.py:902-905)."""
    import torch
    from uhc_amd import sim as S
    from uhc_amd.agents import agent_dict
    from uhc_amd.data_loaders.dataset_amass_single import DatasetAMASSSingle
    from uhc_amd.data_loaders.synthetic import make_synthetic_amass
    from uhc_amd.model.shapes import box_triangles
    from uhc_amd.utils.config_utils.copycat_config import Config
    torch.set_default_dtype(torch.float64)
    cfg = Config(cfg_id="copycat_mi355x", base_dir=str(tmp_path))
    cfg.n_env, cfg.min_batch_size, cfg.num_optim_epoch, cfg.no_log = 64, 64 * 8, 2, True
    cfg.policy_hsize = cfg.value_hsize = [256, 128]
    cfg.save_n_epochs = 1  # checkpoint + eval_policy in the same iteration
    cfg.robot_cfg = {"mesh": True, "model": "smpl", "ball": True}
    cfg.action_type, cfg.residual_force, cfg.meta_pd, cfg.meta_pd_joint = "torque", False, False, False
    cfg.reward_id, cfg.obs_v = "world_rfc_implicit_quat", 2
    cfg.cfg_dict["tq_mul"] = 4
    cfg.env_init_noise = 0.0
    specs = dict(cfg.data_specs)
    specs["file_path"] = "synthetic"
    K = 3
    dl = DatasetAMASSSingle(specs, "train", pickle_data=make_synthetic_amass(6, seed=4, t_range=(40, 80), objects=K))
    s = dl.get_sample_from_key(dl.data_keys[0], full_sample=True)
    assert s["has_obj"] and s["num_obj"] == K and s["obj_pose"].shape[1] == 7 * K
    agent = agent_dict[cfg.agent_name](cfg, torch.float64, torch.device("cuda", 0), data_loader=dl,
                                       objects=dict(hulls=[box_triangles(0.15, 0.15, 0.15)] * K, density=5.0 / 0.027))
    env = agent.env
    assert env.use_quat and env.num_obj == K and (env.model.nq, env.model.nv) == (99 + 7 * K, 75 + 6 * K)
    assert (env.qpos_lim, env.qvel_lim, env.body_lim) == (99, 75, 25) and agent.state_dim == 534 and agent.action_dim == 69
    info = agent.optimize_policy(0)
    log = info["log"]
    assert log.num_steps == 64 * 8 and 0.0 < log.avg_c_reward <= 1.0 and np.isfinite(log.avg_c_info).all()
    assert "log_eval" in info  # eval ran (the fail-safe set_state takes the model's full width)
    assert int(env.sim.field(S.F_FAIL).sum().item()) == 0
    q = env.sim.field(S.F_QPOS).cpu().numpy()
    assert np.abs(np.linalg.norm(q[:, 3:99].reshape(-1, 24, 4), axis=2) - 1).max() < 1e-9
    assert np.isfinite(q).all() and (q[:, 99 + 2::7][:, :K] > -0.05).all()  # the boxes rest on the floor, not under it
    env.close()


def test_facade_env_with_objects(tmp_path):
    """HumanoidEnv (the reference's single-env surface) on a clip with objects: the model gets them (reset_robot), reset puts them at
    obj_pose[0], get_obj_qpos / get_obj_qvel read them, the observation does not grow, fail_safe leaves them where they are."""
    import torch
    from uhc_amd.data_loaders.dataset_amass_single import DatasetAMASSSingle
    from uhc_amd.data_loaders.synthetic import make_synthetic_amass
    from uhc_amd.envs import env_dict
    from uhc_amd.model.shapes import box_triangles
    from uhc_amd.utils.config_utils.copycat_config import Config
    torch.set_default_dtype(torch.float64)
    cfg = Config(cfg_id="copycat_mi355x", base_dir=str(tmp_path))
    cfg.no_log = True
    specs = dict(cfg.data_specs)
    specs["file_path"] = "synthetic"
    dl = DatasetAMASSSingle(specs, "train", pickle_data=make_synthetic_amass(2, seed=4, t_range=(40, 60), objects=2))
    e0 = dl.get_sample_from_key(dl.data_keys[0], full_sample=True)
    e0["obj_mesh"] = [box_triangles(0.15, 0.15, 0.15), box_triangles(0.1, 0.1, 0.1)]
    env = env_dict["humanoid_im"](cfg, init_expert=e0, data_specs=cfg.data_specs, mode="test")
    assert env.observation_space.shape == (657,) and env.model.nq == 76 + 14 and env.vec.num_obj == 2
    obs = env.reset()
    assert obs.shape == (657,) and np.isfinite(obs).all()
    np.testing.assert_allclose(env.get_humanoid_qpos(), env.expert["qpos"][0], atol=1e-12)
    np.testing.assert_allclose(env.get_obj_qpos(), e0["obj_pose"][0], atol=1e-15)
    assert (env.get_obj_qvel() == 0).all() and env.get_wbody_pos().shape == (72,)
    for _ in range(4):
        obs, r, done, info = env.step(np.zeros(env.action_dim))
    assert np.isfinite(obs).all() and np.abs(env.get_obj_qvel()).max() > 0  # the boxes fall
    before = env.get_obj_qpos()
    env.fail_safe()
    np.testing.assert_allclose(env.get_obj_qpos(), before, atol=0)
    np.testing.assert_allclose(env.get_humanoid_qpos(), env.get_expert_qpos(), atol=1e-12)
    vec = env.vec
    e1 = dl.get_sample_from_key(dl.data_keys[1], full_sample=True)
    e1["obj_mesh"] = e0["obj_mesh"]
    env.load_expert(e1)
    assert env.vec is vec  # same body, same objects: nothing rebuilt
    e1["obj_mesh"] = [box_triangles(0.2, 0.1, 0.1), box_triangles(0.1, 0.1, 0.1)]
    env.load_expert(e1)
    assert env.vec is not vec and env.vec.num_obj == 2  # other objects: reset_robot rebuilds the model
    env.vec.close()
