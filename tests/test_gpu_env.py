"""GPU parity of the env layer (PD target gather, termination, reward, observation v2, reset) through the
C-ABI, against the CPU oracles (physics_oracle.c + env_oracle.py) on the same inputs."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REWARD_W = dict(w_p=0.3, w_v=0.1, w_e=0.45, w_c=0.1, w_vf=0.05, k_p=2.0, k_v=0.005, k_e=5.0, k_c=100.0, k_vf=1.0)
REWARD_W23 = dict(k_p=0.6, k_wp=0.3, k_v=0.004, k_j=60.0, k_c=80.0, k_vf=0.7, w_p=0.25, w_wp=0.2, w_v=0.05, w_j=0.3, w_c=0.15, w_vf=0.05,
                  jpos_diffw=[float(x) for x in np.round(np.linspace(0.5, 1.5, 24), 3)])


def _expert():
    f = np.load(os.path.join(G, "g3_qpos_fk.npz"))
    e = {k[2:]: f[k] for k in f.files if k.startswith("f_")}
    e["len"] = int(e["len"])
    return e


def _window(e, start, n, model=None):
    """Expert features of the window [start, start+n) exactly as load_expert computes them: qpos_fk on the window's own
    qpos (so its first-frame velocities are copies of the second frame's finite differences)."""
    import torch
    from uhc_amd.sim import load_asset_model
    from uhc_amd.smpllib.torch_smpl_humanoid import Humanoid
    w = Humanoid(model=model or load_asset_model()).qpos_fk(torch.from_numpy(e["qpos"][start:start + n].copy()))
    return w


def _make(model, ctrl, n_env, expert, beta, obs_v=2, reward_v=0, has_shape=True, env_term_body="body"):
    import torch
    from uhc_amd import sim as S
    from uhc_amd._capi import env_desc
    sb = S.SimBatch(model, ctrl, n_env)
    eb = S.EnvBatch(sb, env_desc(model, obs_v=obs_v, has_shape=has_shape, reward_weights=REWARD_W23 if reward_v >= 4 else REWARD_W, reward_v=reward_v,
                                 fut_frames=3, fut_skip=4, obs_heading=True, root_deheading=True, obs_phase=True, env_term_body=env_term_body))
    frames = S.pack_expert_frames(expert)
    frames2 = np.concatenate([frames, frames[::-1].copy()])  # clip 1 = clip 0 reversed (only a second id to address)
    clip_start = torch.tensor([0, frames.shape[0]], dtype=torch.int32)
    clip_beta = torch.from_numpy(np.stack([np.r_[beta, 2.0], np.r_[beta * 0.5, 1.0]]))
    eb.set_bank(torch.from_numpy(frames2), clip_start, clip_beta)
    return sb, eb


def _oracle_obs(E, obs_v, o, w, t, beta):
    xpos, xquat, xipos = o.get("xpos").reshape(-1, 3), o.get("xquat").reshape(-1, 4), o.get("xipos").reshape(-1, 3)
    if obs_v == 1:
        return E.full_obs_v1(o.get("qpos"), o.get("qvel"), xpos, xquat, xipos, w, t, 0)
    if obs_v == 3:
        return E.full_obs_v3(o.get("qpos"), o.get("qvel"), xpos, xquat, w, t, 0, beta, 2.0, fut_frames=3, skip=4)
    if obs_v == 6:
        return E.full_obs_v6(o.get("qpos"), o.get("qvel"), xpos, w, t, 0, beta, 2.0)
    if obs_v == 5:
        return E.full_obs_v5(o.get("qpos"), o.get("qvel"), xpos, xquat, w, t, 0, beta, 2.0)
    if obs_v == 0:
        return E.full_obs_v0(o.get("qpos"), o.get("qvel"), w, t, 0, obs_heading=True, root_deheading=True, obs_phase=True)
    if obs_v == 4:
        return E.full_obs_v4(o.get("qpos"), o.get("qvel"), xpos, xquat, w, t, 0, beta, 2.0)[0]
    return E.full_obs_v2(o.get("qpos"), o.get("qvel"), xpos, xquat, w, t, 0, beta, 2.0)


@pytest.mark.parametrize("obs_v,reward_v", [(2, 0), (1, 0), (6, 1), (3, 0), (5, 2), (0, 4), (2, 5), (2, 3), (4, 0)])
def test_env_rollout_matches_oracles(model, ctrl, obs_v, reward_v):
    _rollout_check(model, ctrl, obs_v, reward_v)


def test_env_rollout_on_the_generated_model_class(model, ctrl):
    """The env layer (observation, reward, termination) on the model class the reference actually runs: what Robot(cfg.robot_cfg)
    generates -- body-body collisions on (smpl_parser.py:327-328), Chest / shoulder excludes, rel_joint_lm knee / ankle / toe ranges
    (smpl_robot.py:1087-1110) -- instead of the floor-only static asset.  The steps run through the dense-row kernels (and the general
    tier when an env exceeds the fast one); the oracle follows the device's per-substep solver word."""
    import dataclasses
    from uhc_amd.smpllib.smpl_robot import robot_variant
    gen = dataclasses.replace(robot_variant(model, {"mesh": True, "model": "smpl"}), solver=1)
    assert (gen.geom_contype[1:] == 1).all() and gen.nexclude == 2
    assert gen.jnt_range[gen.joint_names.index("L_Knee_x")] == pytest.approx([-np.pi / 16, np.pi])
    stats = _rollout_check(gen, ctrl, 2, 0)
    assert stats["max_ncon_two_body"] > 0  # body-body contacts did occur: the test is not the floor-only one in disguise


def _rollout_check(model, ctrl, obs_v, reward_v):
    import torch
    from oracle import env_oracle as E
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.smpllib.smpl_mujoco import SMPLConverter
    expert = _expert()
    rng = np.random.default_rng(11)
    beta = rng.normal(size=16)
    n = 4
    sb, eb = _make(model, ctrl, n, expert, beta, obs_v, reward_v, has_shape=obs_v != 1)
    starts, lens = np.array([0, 3, 10, 20]), np.array([40, 30, 25, 12])
    ids = torch.arange(n, dtype=torch.int32)
    eb.assign(ids, torch.zeros(n, dtype=torch.int32), torch.from_numpy(starts), torch.from_numpy(lens))
    noise = rng.normal(scale=0.05, size=(n, model.nu))
    eb.reset(ids.cuda(), torch.from_numpy(noise))
    sb.sync()
    jw = SMPLConverter(model, model).get_new_diff_weight()
    wins = [_window(expert, starts[e], lens[e]) for e in range(n)]
    os_ = []
    for e in range(n):
        o = OracleSim(model, ctrl)
        q0 = wins[e]["qpos"][0].copy()
        q0[7:] += noise[e]
        o.set_state(q0, wins[e]["qvel"][0])
        os_.append(o)
    # ---- reset observation
    gobs = eb.field(S.E_OBS).cpu().numpy()
    for e in range(n):
        np.testing.assert_allclose(gobs[e], _oracle_obs(E, obs_v, os_[e], wins[e], 0, beta), atol=1e-11)
    assert eb.obs_dim == {2: 657, 1: 784, 6: 401, 3: 3 * 657, 5: 653, 0: 220, 4: 643}[obs_v]
    # ---- steps
    cur_t = np.zeros(n, dtype=int)
    alive = np.ones(n, dtype=bool)
    two_body = 0
    for t in range(14):
        act = rng.normal(scale=0.1, size=(n, ctrl.action_dim))
        active = torch.from_numpy(alive.astype(np.int32)).cuda()
        eb.step(torch.from_numpy(act).cuda(), active)
        sb.sync()
        gobs, grew = eb.field(S.E_OBS).cpu().numpy(), eb.field(S.E_REWARD).cpu().numpy()
        gdone, gfail, gend = (eb.field(f).cpu().numpy() for f in (S.E_DONE, S.E_FAIL, S.E_END))
        gpct, gparts = eb.field(S.E_PERCENT).cpu().numpy(), eb.field(S.E_REWARD_PARTS).cpu().numpy()
        gq = sb.field(S.F_QPOS).cpu().numpy()
        redo = sb.field(S.F_REDO).cpu().numpy()
        for e in range(n):
            if not alive[e]:
                continue
            o, w = os_[e], wins[e]
            prev_bquat = E.get_body_quat(o.get("qpos"))
            tb = w["qpos"][E.expert_index(cur_t[e] + 1, 0, w["len"])][7:]
            o.do_simulation(act[e], tb, redo=redo[e])  # UHC_F_REDO bits 8+: substeps the general tier solved by sweeps (0 on the fast path)
            two_body = max(two_body, _two_body_contacts(model, o))
            cur_t[e] += 1
            xpos, xquat, xipos = o.get("xpos").reshape(-1, 3), o.get("xquat").reshape(-1, 4), o.get("xipos").reshape(-1, 3)
            np.testing.assert_allclose(gq[e], o.get("qpos"), atol=1e-9)
            bd = E.calc_body_diff(xpos, w["wbpos"][E.expert_index(cur_t[e], 0, w["len"])], jw)
            fail = bool(o.geti("fail")) or bd > 0.5
            end = cur_t[e] >= 100000 or cur_t[e] >= w["len"] - 1
            if reward_v >= 4:
                w23 = {k: v for k, v in REWARD_W23.items() if k != "jpos_diffw"}
                r, parts = E.world_rfc_implicit_v2_v3(reward_v == 5, o.get("qpos"), xpos, xquat, xipos, prev_bquat, act[e], w, cur_t[e], 0, model.timestep * 15,
                                                      w23, np.asarray(REWARD_W23["jpos_diffw"]))
            elif reward_v in (2, 3):
                a = np.r_[act[e][:75], np.zeros(300)] if reward_v == 3 else act[e]
                r, parts = E.world_rfc_mul_reward(reward_v == 3, o.get("qpos"), xpos, xipos, prev_bquat, a, w, cur_t[e], 0, model.timestep * 15, jw[1:], REWARD_W)
            elif reward_v == 1:  # the explicit reward reads 24 x 9 residual entries; this controller's action carries 6 + 30 after the joints
                r, parts = E.world_rfc_explicit_reward(o.get("qpos"), xpos, xipos, prev_bquat, np.r_[act[e][:75], np.zeros(300)], w, cur_t[e], 0,
                                                       model.timestep * 15, jw[1:], REWARD_W)
            else:
                r, parts = E.world_rfc_implicit_reward(o.get("qpos"), xpos, xipos, prev_bquat, act[e], w, cur_t[e], 0, model.timestep * 15, jw[1:], REWARD_W)
            obs = _oracle_obs(E, obs_v, o, w, cur_t[e], beta)
            assert (bool(gfail[e]), bool(gend[e]), bool(gdone[e])) == (fail, end, fail or end)
            assert gpct[e] == pytest.approx(cur_t[e] / (w["len"] - 1), abs=1e-14)
            assert grew[e] == pytest.approx(r, abs=1e-9)
            np.testing.assert_allclose(gparts[e][:len(parts)], parts, atol=1e-9)
            np.testing.assert_allclose(gobs[e], obs, atol=1e-8)
            if fail or end:
                alive[e] = False
    assert not alive[3]  # the 12-frame window must have ended
    return {"max_ncon_two_body": two_body}


def _two_body_contacts(model, o):
    """Contacts of the oracle's last forward pass that are not with the floor (ncon minus the plane contacts, by height of the normal)."""
    n = o.geti("ncon")
    if n == 0:
        return 0
    fr = o.get("con_frame").reshape(-1, 9)[:n]
    pos = o.get("con_pos").reshape(-1, 3)[:n]
    return int(((np.abs(fr[:, 2]) < 0.999) | (np.abs(pos[:, 2]) > 0.02)).sum())


def test_env_inactive_and_second_clip(model, ctrl):
    import torch
    from uhc_amd import sim as S
    expert = _expert()
    beta = np.linspace(-1, 1, 16)
    n = 3
    sb, eb = _make(model, ctrl, n, expert, beta)
    ids = torch.arange(n, dtype=torch.int32)
    eb.assign(ids, torch.tensor([0, 1, 1], dtype=torch.int32), torch.tensor([0, 0, 5], dtype=torch.int32), torch.tensor([40, 40, 20], dtype=torch.int32))
    eb.reset(ids.cuda(), None)
    sb.sync()
    obs = eb.field(S.E_OBS).cpu().numpy()
    np.testing.assert_allclose(obs[0, 640:656], beta)
    np.testing.assert_allclose(obs[1, 640:657], np.r_[beta * 0.5, 1.0])
    q = sb.field(S.F_QPOS).cpu().numpy()
    np.testing.assert_allclose(q[1], expert["qpos"][39], atol=1e-15)   # clip 1 is clip 0 reversed
    np.testing.assert_allclose(q[2], expert["qpos"][34], atol=1e-15)
    before = eb.field(S.E_OBS).clone()
    act = torch.zeros(n, ctrl.action_dim, dtype=torch.float64, device="cuda")
    eb.step(act, torch.tensor([0, 1, 0], dtype=torch.int32, device="cuda"))
    sb.sync()
    after = eb.field(S.E_OBS)
    assert torch.equal(before[0], after[0]) and torch.equal(before[2], after[2]) and not torch.equal(before[1], after[1])
    assert eb.field(S.E_CUR_T).cpu().tolist() == [0, 1, 0]


def test_env_term_body_root(model, ctrl):
    """cfg.env_term_body == "root" (humanoid_im.py:1225-1226): an env fails when its root is more than 0.1 below the lowest root height of
    its expert window -- also for a window taken through the device-side queue."""
    import torch
    from uhc_amd import sim as S
    expert = dict(_expert())
    T = expert["qpos"].shape[0]
    expert["qpos"] = expert["qpos"].copy()
    expert["qpos"][:, 2] += 0.3 * np.arange(T) / (T - 1)  # the expert root rises by 0.3 over the clip
    n = 3
    sb, eb = _make(model, ctrl, n, expert, np.zeros(16), env_term_body="root")
    ids = torch.arange(n, dtype=torch.int32)
    # env 0 tracks the whole clip (lowest height = frame 0), env 1 its upper half, env 2 a 2-frame window with the upper half queued behind it
    eb.assign(ids, torch.zeros(n, dtype=torch.int32), torch.tensor([0, 20, 0], dtype=torch.int32), torch.tensor([T, T - 20, 2], dtype=torch.int32))
    eb.set_next(torch.tensor([2], dtype=torch.int32), torch.tensor([0], dtype=torch.int32), torch.tensor([20], dtype=torch.int32), torch.tensor([T - 20], dtype=torch.int32), None)
    eb.reset(ids.cuda(), None)
    # all three stand at the height of frame 0, i.e. 0.154 below the lowest frame of the upper half
    q0 = np.tile(_expert()["qpos"][0], (n, 1))
    sb.set_state(torch.from_numpy(q0), torch.zeros(n, model.nv, dtype=torch.float64))
    act = torch.zeros(n, ctrl.action_dim, dtype=torch.float64, device="cuda")
    eb.step(act, None)
    sb.sync()
    assert eb.field(S.E_FAIL).cpu().tolist() == [0, 1, 0] and eb.field(S.E_DONE).cpu().tolist() == [0, 1, 1]  # env 2: end of its 2-frame window
    eb.auto_reset()   # env 1 restarts its window, env 2 takes the queued upper half: both start at that window's own height
    eb.step(act, None)
    sb.sync()
    assert eb.field(S.E_FAIL).cpu().tolist() == [0, 0, 0]
    sb.set_state(torch.from_numpy(q0), torch.zeros(n, model.nv, dtype=torch.float64))
    eb.step(act, None)
    sb.sync()
    assert eb.field(S.E_FAIL).cpu().tolist() == [0, 1, 1]   # the queued window's lower bound applies to env 2 now
    sb.close()


def test_auto_reset_equals_assign_plus_reset(model, ctrl):
    """Device-side episode turnover: a finished env restarts from its queued window (+ noise) exactly as assign + reset
    would restart it; an env without a queued window restarts its current one; running envs are untouched."""
    import torch
    from uhc_amd import sim as S
    expert = _expert()
    beta = np.linspace(-1, 1, 16)
    n = 4
    sb, eb = _make(model, ctrl, n, expert, beta)
    ids = torch.arange(n, dtype=torch.int32)
    # envs 0 and 1 track 3-frame windows (done after 2 steps), envs 2 and 3 long ones
    eb.assign(ids, torch.zeros(n, dtype=torch.int32), torch.tensor([0, 4, 0, 8], dtype=torch.int32), torch.tensor([3, 3, 40, 30], dtype=torch.int32))
    eb.reset(ids.cuda(), None)
    rng = np.random.default_rng(3)
    noise = torch.from_numpy(rng.normal(scale=0.05, size=(2, model.nu)))
    eb.set_next(torch.tensor([0, 2], dtype=torch.int32), torch.tensor([1, 1], dtype=torch.int32), torch.tensor([6, 2], dtype=torch.int32),
                torch.tensor([20, 25], dtype=torch.int32), noise)
    act = torch.zeros(n, ctrl.action_dim, dtype=torch.float64, device="cuda")
    for _ in range(2):
        eb.step(act, None)
    sb.sync()
    assert eb.field(S.E_DONE).cpu().tolist() == [1, 1, 0, 0]
    keep_obs, keep_q = eb.field(S.E_OBS).clone(), sb.field(S.F_QPOS).clone()
    eb.auto_reset()
    sb.sync()
    assert eb.field(S.E_CONSUMED).cpu().tolist() == [1, 0, 0, 0] and eb.field(S.E_CUR_T).cpu().tolist() == [0, 0, 2, 2]
    auto_obs, auto_q, auto_v = eb.field(S.E_OBS).clone(), sb.field(S.F_QPOS).clone(), sb.field(S.F_QVEL).clone()
    assert torch.equal(auto_obs[2:], keep_obs[2:]) and torch.equal(auto_q[2:], keep_q[2:])  # running envs untouched
    # the same restarts through the explicit path
    eb.assign(torch.tensor([0], dtype=torch.int32), torch.tensor([1], dtype=torch.int32), torch.tensor([6], dtype=torch.int32), torch.tensor([20], dtype=torch.int32))
    eb.reset(torch.tensor([0], dtype=torch.int32).cuda(), noise[0:1])
    eb.reset(torch.tensor([1], dtype=torch.int32).cuda(), None)  # env 1 had nothing queued: same window again, no noise
    sb.sync()
    assert torch.equal(eb.field(S.E_OBS)[:2], auto_obs[:2]) and torch.equal(sb.field(S.F_QPOS)[:2], auto_q[:2]) and torch.equal(sb.field(S.F_QVEL)[:2], auto_v[:2])
    # env 2's queued window is still waiting; it is taken when env 2 finishes
    for _ in range(37):
        eb.step(act, torch.tensor([0, 0, 1, 0], dtype=torch.int32, device="cuda"))
    sb.sync()
    assert eb.field(S.E_DONE).cpu().tolist()[2] == 1
    eb.auto_reset()
    sb.sync()
    assert eb.field(S.E_CONSUMED).cpu().tolist()[2] == 1
    q2 = sb.field(S.F_QPOS)[2].cpu().numpy()
    np.testing.assert_allclose(q2[:7], expert["qpos"][39 - 2][:7], atol=1e-15)  # clip 1 = clip 0 reversed, window start 2


def test_per_clip_body_shape_switches_the_env_model(tmp_path):
    """smpl_shape-style operation: every clip carries its own body shape (model); an env runs on the model of the clip it
    is assigned -- at assign + reset and at a device-side restart -- and matches the oracle built on that model."""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.data_loaders.synthetic import make_synthetic_amass
    from uhc_amd.envs.humanoid_im import VecHumanoidEnv
    from uhc_amd.model.mjcf import scale_model
    from uhc_amd.utils.config_utils.copycat_config import Config
    cfg = Config(cfg_id="copycat_mi355x", base_dir=str(tmp_path))
    cfg.env_init_noise = 0.0
    base = S.load_asset_model()
    tall = scale_model(base, 1.12)
    env = VecHumanoidEnv(cfg, n_env=3, shape_models=[tall])
    pk = make_synthetic_amass(2, seed=5, t_range=(40, 50))
    keys = list(pk.keys())
    clips = {k: dict(pose_aa=v["pose_aa"], trans=v["trans"], beta=np.zeros(16), gender=0) for k, v in pk.items()}
    env.set_clip_bank(clips, clip_model={keys[0]: 0, keys[1]: 1})
    env.assign(np.arange(3), [keys[0], keys[1], keys[1]], [0, 0, 5], [30, 30, 8])
    env.reset(np.arange(3))
    env.sim.sync()
    feats = [env.expert_features(clips[keys[0]], 0), env.expert_features(clips[keys[1]], 1)]
    models = env.models  # the env's copies carry the configured PGS sweep cap
    xpos = env.sim.field(S.F_XPOS).cpu().numpy()
    os_ = []
    for e, (ci, st) in enumerate([(0, 0), (1, 0), (1, 5)]):
        o = OracleSim(models[ci], env.ctrl)
        o.set_state(feats[ci]["qpos"][st], feats[ci]["qvel"][st + 1])
        np.testing.assert_allclose(xpos[e], o.get("xpos"), atol=1e-12)
        os_.append(o)
    assert np.abs(xpos[1] - OracleSim(base, env.ctrl).get("xpos")[: xpos.shape[1]]).max() > 1e-3  # really another body
    # a few control steps on the respective models
    act = np.random.default_rng(2).normal(scale=0.05, size=(3, env.action_dim))
    for t in range(2):
        env.step(torch.from_numpy(act).cuda())
        env.sim.sync()
        gq = env.sim.field(S.F_QPOS).cpu().numpy()
        redo = env.sim.field(S.F_REDO).cpu().numpy()
        for e, (ci, st) in enumerate([(0, 0), (1, 0), (1, 5)]):
            os_[e].do_simulation(act[e], feats[ci]["qpos"][min(st + t + 1, st + [30, 30, 3][e] - 1)][7:], redo=redo[e])  # UHC_F_REDO bits 8+: the substeps the general kernel solved by sweeps
            np.testing.assert_allclose(gq[e], os_[e].get("qpos"), atol=1e-9)
    # env 2's 3-frame window is over: queue a window of the OTHER clip and restart on the device
    assert env.done.cpu().tolist() == [0, 0, 1]
    env.set_next([2], [keys[0]], [3], [25])
    env.auto_reset()
    env.sim.sync()
    o = OracleSim(models[0], env.ctrl)
    o.set_state(feats[0]["qpos"][3], feats[0]["qvel"][4])
    np.testing.assert_allclose(env.sim.field(S.F_XPOS)[2].cpu().numpy(), o.get("xpos"), atol=1e-12)
    env.close()


@pytest.mark.parametrize("solver", [1, 0])
def test_deferred_reset_forward_matches_explicit_reset(model, ctrl, solver):
    """uhc_env_auto_reset only refreshes the kinematics of a restarted env and leaves the rest of sim.forward() to the head of the
    env's next step kernel: the steps that follow are bit-identical to those after an explicit assign + reset (full forward pass
    at once), for envs restarted from a queued window and for envs restarting their own window."""
    import dataclasses
    import torch
    from uhc_amd import sim as S
    model = dataclasses.replace(model, solver=solver)
    expert = _expert()
    beta = np.linspace(-1, 1, 16)
    n = 4
    rng = np.random.default_rng(8)
    noise = torch.from_numpy(rng.normal(scale=0.05, size=(2, model.nu)))
    acts = [torch.from_numpy(rng.normal(scale=0.1, size=(n, ctrl.action_dim))).cuda() for _ in range(5)]
    ids = torch.arange(n, dtype=torch.int32)
    out = []
    for path in ("auto", "explicit"):
        sb, eb = _make(model, ctrl, n, expert, beta)
        eb.assign(ids, torch.zeros(n, dtype=torch.int32), torch.tensor([0, 4, 0, 8], dtype=torch.int32), torch.tensor([3, 3, 40, 30], dtype=torch.int32))
        eb.reset(ids.cuda(), None)
        if path == "auto":
            eb.set_next(torch.tensor([0], dtype=torch.int32), torch.tensor([1], dtype=torch.int32), torch.tensor([6], dtype=torch.int32), torch.tensor([20], dtype=torch.int32), noise[0:1])
        for t in range(2):
            eb.step(acts[t], None)
        sb.sync()
        assert eb.field(S.E_DONE).cpu().tolist() == [1, 1, 0, 0]
        if path == "auto":
            eb.auto_reset()
        else:
            eb.assign(torch.tensor([0], dtype=torch.int32), torch.tensor([1], dtype=torch.int32), torch.tensor([6], dtype=torch.int32), torch.tensor([20], dtype=torch.int32))
            eb.reset(torch.tensor([0], dtype=torch.int32).cuda(), noise[0:1])
            eb.reset(torch.tensor([1], dtype=torch.int32).cuda(), None)
        rec = [eb.field(S.E_OBS).clone()]
        for t in range(2, 5):
            eb.step(acts[t], None)
            rec += [eb.field(S.E_OBS).clone(), sb.field(S.F_QPOS).clone(), sb.field(S.F_QVEL).clone(), eb.field(S.E_REWARD).clone()]
        sb.sync()
        rec += [sb.field(S.F_QM).clone(), sb.field(S.F_QFRC_BIAS).clone()]
        out.append(rec)
        sb.close()
    for a, b in zip(*out):
        assert torch.equal(a, b)


def test_ball_env_rollout_matches_oracles(model):
    """The ball-joint env (robot.ball / `use_quat`, config/copycat_ball: torque actions, no residual force, reward world_rfc_implicit_quat,
    observation get_full_obs_v2_quat): reset state = the expert's quaternion pose, body quaternions straight out of qpos, the 534-wide
    observation -- device kernels vs env_oracle (pinned by the reference fixture G14) + the physics oracle on the ball model."""
    import dataclasses
    import torch
    from oracle import env_oracle as E
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd._capi import env_desc
    from uhc_amd.model.mjcf import ball_variant
    from uhc_amd.smpllib.smpl_mujoco import SMPLConverter, smpl_to_qpose
    from uhc_amd.smpllib.torch_smpl_humanoid import Humanoid
    g = np.load(os.path.join(G, "g14_ball_env.npz"))
    ball = dataclasses.replace(ball_variant(model), solver=1)
    ctrl = S.make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
    jw = SMPLConverter(model, model).get_new_diff_weight()
    n = 3
    sb = S.SimBatch(ball, ctrl, n)
    eb = S.EnvBatch(sb, env_desc(model, obs_v=2, has_shape=True, reward_weights=REWARD_W, reward_v=0, jpos_diffw=jw))
    assert eb.obs_dim == 534
    qpos_q = g["qpos_quat"]
    starts, lens = np.array([0, 4, 12]), np.array([30, 20, 9])

    def window(s, l):
        w = Humanoid(model=model).qpos_fk(torch.from_numpy(g["qpos_euler"][s:s + l].copy()))
        w["qpos_quat"] = qpos_q[s:s + l]
        return w

    whole = Humanoid(model=model).qpos_fk(torch.from_numpy(g["qpos_euler"].copy()))
    rec = np.array(whole["qpos"], copy=True)
    rec[:, 3:7] = qpos_q[:, 3:7]
    frames = S.pack_expert_frames(dict(whole, qpos=rec))
    beta = g["c0_beta"]
    eb.set_bank(torch.from_numpy(frames), torch.tensor([0], dtype=torch.int32), torch.from_numpy(np.r_[beta, 2.0][None]))
    ids = torch.arange(n, dtype=torch.int32)
    eb.assign(ids, torch.zeros(n, dtype=torch.int32), torch.from_numpy(starts), torch.from_numpy(lens))
    eb.reset(ids.cuda(), None)
    sb.sync()
    wins = [window(starts[e], lens[e]) for e in range(n)]
    os_ = []
    gq0 = sb.field(S.F_QPOS).cpu().numpy()
    for e in range(n):
        o = OracleSim(ball, ctrl)
        o.set_state(wins[e]["qpos_quat"][0], wins[e]["qvel"][0])
        np.testing.assert_allclose(gq0[e], wins[e]["qpos_quat"][0], atol=1e-15)  # reset to the quaternion expert pose
        os_.append(o)
    gobs = eb.field(S.E_OBS).cpu().numpy()
    for e in range(n):
        xpos, xquat = os_[e].get("xpos").reshape(-1, 3), os_[e].get("xquat").reshape(-1, 4)
        np.testing.assert_allclose(gobs[e], E.full_obs_v2_quat(os_[e].get("qpos"), os_[e].get("qvel"), xpos, xquat, wins[e], 0, 0, beta, 2.0), atol=1e-11)
    rng = np.random.default_rng(17)
    cur_t, alive = np.zeros(n, dtype=int), np.ones(n, dtype=bool)
    for t in range(10):
        act = rng.normal(scale=0.003, size=(n, ctrl.action_dim))
        eb.step(torch.from_numpy(act).cuda(), torch.from_numpy(alive.astype(np.int32)).cuda())
        sb.sync()
        gobs, grew, gparts = eb.field(S.E_OBS).cpu().numpy(), eb.field(S.E_REWARD).cpu().numpy(), eb.field(S.E_REWARD_PARTS).cpu().numpy()
        gdone, gq, redo = eb.field(S.E_DONE).cpu().numpy(), sb.field(S.F_QPOS).cpu().numpy(), sb.field(S.F_REDO).cpu().numpy()
        for e in range(n):
            if not alive[e]:
                continue
            o, w = os_[e], wins[e]
            prev_bquat = E.get_body_quat_ball(o.get("qpos"))
            o.do_simulation(act[e], np.zeros(69), redo=redo[e])
            cur_t[e] += 1
            xpos, xquat, xipos = o.get("xpos").reshape(-1, 3), o.get("xquat").reshape(-1, 4), o.get("xipos").reshape(-1, 3)
            np.testing.assert_allclose(gq[e], o.get("qpos"), atol=1e-9)
            r, parts = E.world_rfc_implicit_reward(o.get("qpos"), xpos, xipos, prev_bquat, act[e], w, cur_t[e], 0, model.timestep * 15, jw[1:], REWARD_W, vf_dim=0, ball=True)
            bd = E.calc_body_diff(xpos, w["wbpos"][E.expert_index(cur_t[e], 0, w["len"])], jw)
            fail, end = bool(o.geti("fail")) or bd > 0.5, cur_t[e] >= w["len"] - 1
            assert bool(gdone[e]) == (fail or end)
            assert grew[e] == pytest.approx(r, abs=1e-9) and parts[4] == 0.0
            np.testing.assert_allclose(gparts[e][:5], parts, atol=1e-9)
            np.testing.assert_allclose(gobs[e], E.full_obs_v2_quat(o.get("qpos"), o.get("qvel"), xpos, xquat, w, cur_t[e], 0, beta, 2.0), atol=1e-8)
            if fail or end:
                alive[e] = False
    assert not alive[2]  # the 9-frame window has ended
