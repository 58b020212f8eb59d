"""GPU end-to-end: AgentCopycat on synthetic clips -- batched rollout, GAE, PPO update, checkpoints, facade env."""
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg(tmp_path, n_env=64, batch=64 * 8):
    from uhc_amd.utils.config_utils.copycat_config import Config
    cfg = Config(cfg_id="copycat_mi355x", base_dir=str(tmp_path))
    cfg.n_env, cfg.min_batch_size, cfg.num_optim_epoch, cfg.no_log = n_env, batch, 2, True
    cfg.policy_hsize = cfg.value_hsize = [256, 128]
    cfg.save_n_epochs = 1
    return cfg


def _loader(cfg, n=6):
    from uhc_amd.data_loaders.dataset_amass_single import DatasetAMASSSingle
    from uhc_amd.data_loaders.synthetic import make_synthetic_amass
    specs = dict(cfg.data_specs)
    specs["file_path"] = "synthetic"
    return DatasetAMASSSingle(specs, "train", pickle_data=make_synthetic_amass(n, seed=4, t_range=(40, 80)))


def test_agent_iteration_and_checkpoint_roundtrip(tmp_path):
    import torch
    from uhc_amd.agents import agent_dict
    torch.set_default_dtype(torch.float64)
    cfg = _cfg(tmp_path)
    agent = agent_dict[cfg.agent_name](cfg, torch.float64, torch.device("cuda", 0), data_loader=_loader(cfg))
    assert agent.state_dim == 657 and agent.action_dim == 105
    before = {k: v.clone() for k, v in agent.policy_net.state_dict().items()}
    info = agent.optimize_policy(0)
    log = info["log"]
    assert log.num_steps == 64 * 8 and 0.0 < log.avg_c_reward <= 1.0 and np.isfinite(log.avg_c_info).all()
    # reset observation + 8 steps per env, + one reset observation per evaluated clip (eval runs after the checkpoint is
    # written and filters its first observation with update=True, agent_copycat.py:445-446)
    # ... + the last observation of every episode that finished inside the pass (agent.py:77-79 filters it too)
    assert agent.running_state.rs.n == 64 * 9 + 6 + log.num_episodes
    changed = [k for k, v in agent.policy_net.state_dict().items() if not torch.equal(v, before[k])]
    assert "action_mean.weight" in changed and "action_log_std" not in changed  # fix_std
    # standing clips + noise actions: nothing blows up
    assert int(agent.env.sim.field(11).sum().item()) == 0
    path = os.path.join(cfg.model_dir, "iter_0001.p")
    assert os.path.exists(path)
    cp = pickle.load(open(path, "rb"))
    assert set(cp.keys()) == {"policy_dict", "value_dict", "running_state"}
    assert cp["policy_dict"]["net.affine_layers.0.weight"].device.type == "cpu" and cp["policy_dict"]["net.affine_layers.0.weight"].dtype == torch.float64
    w = agent.policy_net.action_mean.weight.detach().clone()
    agent.policy_net.action_mean.weight.data.zero_()
    agent.load_checkpoint(1)
    assert torch.equal(agent.policy_net.action_mean.weight.detach(), w)
    agent.env.close()


def test_single_env_facade_matches_batched_env(tmp_path):
    """HumanoidEnv(cfg, init_expert, ...) -- the reference's single-env surface -- drives the same kernels."""
    import torch
    from uhc_amd.envs import env_dict
    torch.set_default_dtype(torch.float64)
    cfg = _cfg(tmp_path)
    dl = _loader(cfg)
    np.random.seed(1)
    expert = dl.sample_seq()
    env = env_dict["humanoid_im"](cfg, init_expert=expert, data_specs=cfg.data_specs, mode="test")
    assert env.observation_space.shape == (657,) and env.action_space.shape == (105,)
    obs = env.reset()
    assert obs.shape == (657,) and np.isfinite(obs).all()
    np.testing.assert_allclose(env.data.qpos, env.expert["qpos"][0], atol=1e-12)  # test mode: no init noise
    total = 0.0
    for t in range(5):
        obs, r, done, info = env.step(np.zeros(105))
        assert r == 1.0 and set(info) == {"fail", "end", "percent"}  # env-native reward is the constant 1 (humanoid_im.py:1222)
        rew, parts = env.last_reward
        total += rew
        assert info["percent"] == pytest.approx((t + 1) / (env.expert["len"] - 1))
    assert env.cur_t == 5 and 0 < total <= 5 and not done
    env.vec.close()


def test_single_env_facade_hands_out_the_obs_v4_tuple(tmp_path):
    """obs_v 4 (uhc/envs/humanoid_im.py:769-861; config/smpl_shape/copycat_disc_1.yml, config/bigfoot/bigfoot_10.yml): the env's observation is the reference's
    obs_full, get_full_obs_v4() returns the reference's tuple (obs_full, local_obs, global_obs), and an agent iteration runs on it."""
    import torch
    from uhc_amd.agents import agent_dict
    from uhc_amd.envs import env_dict
    torch.set_default_dtype(torch.float64)
    cfg = _cfg(tmp_path, n_env=8, batch=8 * 4)
    cfg.obs_v = 4
    dl = _loader(cfg)
    np.random.seed(1)
    env = env_dict["humanoid_im"](cfg, init_expert=dl.sample_seq(), data_specs=cfg.data_specs, mode="test")
    assert env.observation_space.shape == (643,)
    obs = env.reset()
    obs, _, _, _ = env.step(np.zeros(105))
    full, local, glob = env.get_full_obs_v4()
    assert full.shape == (643,) and local.shape == (23, 26) and glob.shape == (45,)
    np.testing.assert_array_equal(full, obs)
    np.testing.assert_array_equal(np.concatenate([glob, local.ravel()]), full)
    np.testing.assert_allclose(np.linalg.norm(local[:, 18:22], axis=1), 1.0, atol=1e-9)  # de-headed world quaternions of the 23 non-root bodies
    np.testing.assert_allclose(local[:, 6:9], local[:, 0:3] - local[:, 3:6], atol=1e-14)  # joint-angle differences
    env.vec.close()
    agent = agent_dict[cfg.agent_name](cfg, torch.float64, torch.device("cuda", 0), data_loader=_loader(cfg, n=5))
    agent.optimize_policy(0, save_model=False)
    assert agent.env.obs_dim == 643
    agent.env.close()


def test_eval_policy_reports_reference_metrics(tmp_path):
    import torch
    from uhc_amd.agents import agent_dict
    torch.set_default_dtype(torch.float64)
    cfg = _cfg(tmp_path, n_env=8, batch=8 * 4)
    agent = agent_dict[cfg.agent_name](cfg, torch.float64, torch.device("cuda", 0), data_loader=_loader(cfg, n=5))
    agent.optimize_policy(0, save_model=False)
    out = agent.eval_policy(0)
    assert len(out) == 1
    (name, m), = out[0].items()
    assert name == "coverage_synthetic" and m["all_coverage"] == 5
    for k in ("mpjpe", "mpjpe_g", "accel_dist", "vel_dist", "root_dist", "reward", "mean_coverage"):
        assert np.isfinite(m[k]), k
    assert 0.0 <= m["mean_coverage"] <= 1.0 and m["mpjpe"] >= 0
    cov = agent.eval_seqs(agent.data_loader.data_keys[:2], agent.data_loader)
    for k, res in cov.items():
        T = agent.data_loader.get_sample_len_from_key(k)
        assert res["pred"].shape[1] == 76 and 2 <= res["pred"].shape[0] == res["gt"].shape[0] and (res["fail_safe"] or res["pred"].shape[0] <= T + 1)
        assert res["pred_jpos"].shape[1] == 72 and res["succ"].shape == (1,)
    agent.env.close()


def test_release_implicit_variant_mcp_obs_v1(tmp_path):
    """The `uhc_implicit` release shape: observation v1 (784), multiplicative-compositional actor, implicit RFC without
    meta-PD (action 75); one training iteration end to end on the device."""
    import torch
    from uhc_amd.agents import agent_dict
    from uhc_amd.models.policy_mcp import PolicyMCP
    torch.set_default_dtype(torch.float64)
    cfg = _cfg(tmp_path, n_env=32, batch=32 * 6)
    cfg.obs_v, cfg.has_shape, cfg.meta_pd, cfg.actor_type, cfg.num_primitive = 1, False, False, "mcp", 8
    cfg.cfg_dict["composer_dim"] = [64, 32]
    cfg.save_n_epochs = 100
    agent = agent_dict[cfg.agent_name](cfg, torch.float64, torch.device("cuda", 0), data_loader=_loader(cfg))
    assert agent.state_dim == 784 and agent.action_dim == 75 and isinstance(agent.policy_net, PolicyMCP)
    before = agent.policy_net.nets[3][1].weight.detach().clone()
    info = agent.optimize_policy(0)
    assert info["log"].num_steps == 32 * 6 and np.isfinite(info["log"].avg_c_info).all()
    assert not torch.equal(before, agent.policy_net.nets[3][1].weight)
    agent.env.close()


def test_release_explicit_variant(tmp_path):
    """The `uhc_explicit` release shape: observation v2 + shape, meta-PD, explicit per-body residual forces
    (action 69 + 24*9 + 30 = 315) and the explicit reward; one training iteration on the device."""
    import torch
    from uhc_amd.agents import agent_dict
    torch.set_default_dtype(torch.float64)
    cfg = _cfg(tmp_path, n_env=32, batch=32 * 6)
    cfg.residual_force_mode, cfg.reward_id = "explicit", "world_rfc_explicit"
    cfg.save_n_epochs = 100
    agent = agent_dict[cfg.agent_name](cfg, torch.float64, torch.device("cuda", 0), data_loader=_loader(cfg))
    assert agent.state_dim == 657 and agent.action_dim == 315 and agent.env.vf_dim == 216
    info = agent.optimize_policy(0)
    assert info["log"].num_steps == 32 * 6 and np.isfinite(info["log"].avg_c_info).all()
    agent.env.close()


def test_split_k_linear_gradients():
    """The full-batch learner's Linear layers take a split-K weight-gradient path on the GPU: same gradients as autograd's."""
    import torch
    from uhc_amd.khrylib.models.mlp import MLP
    from uhc_amd.khrylib.rl.core import Value
    torch.manual_seed(0)
    net = Value(MLP(37, (64, 48), "gelu")).double().cuda()
    x = torch.randn(4096 + 8, 37, dtype=torch.float64, device="cuda")      # not divisible by 32 -> plain path for the head
    x2 = torch.randn(8192, 37, dtype=torch.float64, device="cuda")
    for xx in (x, x2):
        net.zero_grad()
        net(xx).pow(2).mean().backward()
        got = [p.grad.clone() for p in net.parameters()]
        net.zero_grad()
        h = xx
        for l in net.net.affine_layers:
            h = torch.nn.functional.gelu(torch.nn.functional.linear(h, l.weight, l.bias))
        torch.nn.functional.linear(h, net.value_head.weight, net.value_head.bias).pow(2).mean().backward()
        for g, p in zip(got, net.parameters()):
            assert torch.allclose(g, p.grad, rtol=1e-11, atol=1e-14)


def test_fit_uhc_single_clip_loop(tmp_path):
    """scripts/fit_uhc.py: while the current clip is not tracked to its end every sampled window comes from that clip and
    iter_best.p is rewritten; a tracked clip is stored under models_singles/ and the loop moves on."""
    import importlib.util
    import torch
    from uhc_amd.agents import agent_dict
    torch.set_default_dtype(torch.float64)
    cfg = _cfg(tmp_path, n_env=8, batch=8 * 3)
    agent = agent_dict[cfg.agent_name](cfg, torch.float64, torch.device("cuda", 0), data_loader=_loader(cfg, n=3))
    agent.precision_mode = True
    agent.save_curr()
    spec = importlib.util.spec_from_file_location("fit_uhc", os.path.join(os.path.dirname(__file__), "..", "scripts", "fit_uhc.py"))
    fit_uhc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fit_uhc)
    keys = agent.data_loader.data_keys
    seen = []
    orig = agent._sample_windows
    agent._sample_windows = lambda n: (seen.append(orig(n)[0]), orig(n))[1]
    # a random policy does not track a clip to its end: two epochs of fitting on the first clip
    lines = []
    fitted = fit_uhc.fit(agent, 0, 2, log=lines.append)
    assert fitted == {} and len(lines) == 2 and all(ln.startswith(f"Fitting: {keys[0]}") for ln in lines)
    assert seen and all(set(ks) == {keys[0]} for ks in seen)
    assert os.path.exists(f"{cfg.model_dir}/iter_best.p") and agent.fit_single_key == ""
    # with success forced the loop stores every clip's weights and finishes
    agent.eval_seq = lambda k, loader: {"succ": np.array([True])}
    fitted = fit_uhc.fit(agent, 2, 10, log=lines.append)
    assert list(fitted) == list(keys)
    assert sorted(os.listdir(f"{cfg.model_dir}_singles")) == sorted(f"{k}.p" for k in keys)
    assert fit_uhc.fit(agent, 0, 3, log=lines.append) == {}  # everything is done: nothing left to fit
    agent.env.close()


def test_episodes_continue_across_sampling_passes(tmp_path):
    """An episode cut by the end of a sampling pass goes on in the next pass (the reference runs every episode to fail / end): the
    env's frame counter keeps counting, long windows get finished and recorded, and the logged reward is the plain imitation reward."""
    import torch
    from uhc_amd.agents import agent_dict
    from uhc_amd import sim as S
    torch.set_default_dtype(torch.float64)
    cfg = _cfg(tmp_path, n_env=16, batch=16 * 5)
    cfg.save_n_epochs = 1000
    agent = agent_dict[cfg.agent_name](cfg, torch.float64, torch.device("cuda", 0), data_loader=_loader(cfg))
    agent.per_epoch_update(0)
    agent.noise_rate = 0.0
    agent.rollout_begin(5)
    for _ in range(5):
        agent.rollout_step()
    t1 = agent.env.cur_t.cpu().numpy().copy()
    plain = float(agent._ro.c_reward_sum.item()) / (16 * 5)
    agent.value_net.value_head.bias.data.fill_(3.0)  # make the bootstrap visible: V(s_T) ~ 3
    batch, log = agent.rollout_end()
    assert float(batch.rewards.mean().item()) > plain + 0.1  # the bootstrap went into the samples ...
    still = np.nonzero(batch.masks.reshape(16, 5)[:, :4].cpu().numpy().min(1) > 0)[0]  # envs whose first episode was still running
    assert len(still) > 0 and (t1[still] == 5).all()
    # the logged reward is the mean imitation reward: the gamma * V bootstrap folded into the last sample is not in it
    assert log.avg_c_reward == pytest.approx(plain, rel=1e-12) and 0.0 < plain <= 1.0  # ... but not into the logged reward
    n_before = agent.running_state.rs.n
    agent.rollout_begin(5)  # second pass: continues
    assert (agent.env.cur_t.cpu().numpy()[still] == 5).all()
    assert agent.running_state.rs.n == n_before  # the observations the envs stopped at are not counted twice
    agent.rollout_step()
    t2 = agent.env.cur_t.cpu().numpy()
    done = agent.env.done.cpu().numpy().astype(bool)
    assert ((t2[still] == 6) | done[still] | (t2[still] == 0)).all() and (t2[still] == 6).any()
    for _ in range(4):
        agent.rollout_step()
    agent.rollout_end()
    agent.rollout_begin(3, fresh=True)  # an explicit restart assigns and resets every env
    assert (agent.env.cur_t.cpu().numpy() == 0).all()
    agent.env.close()


def test_train_script_single_env_plumbing(tmp_path):
    """BASELINE configs[0]: scripts/train_uhc.py end to end with ONE environment (the reference's shape: 1 env, python loop), through
    the reference's import paths (`uhc.agents`, `uhc.utils.config_utils`)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "train_uhc.py"), "--cfg", "copycat_mi355x", "--synthetic", "4", "--num_epoch", "2",
                          "--n_env", "1", "--min_batch_size", "64", "--no_log"], cwd=str(tmp_path), capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, PYTHONPATH=root))
    assert out.returncode == 0, out.stderr[-3000:]
    assert "training done!" in out.stdout
    assert out.stderr.count("Ep: ") + out.stdout.count("Ep: ") >= 2


def test_facade_accessors_match_device_fields(tmp_path):
    """get_ee_pos / get_body_quat / get_com / fail_safe / prev_bquat of the single-env facade (humanoid_im.py:902-965) against the
    device state they are read from."""
    import torch
    from uhc_amd.envs import env_dict
    from uhc_amd.smpllib.smpl_mujoco import SMPL_EE_NAMES
    from uhc_amd.utils.transformation import quaternion_from_euler
    torch.set_default_dtype(torch.float64)
    cfg = _cfg(tmp_path, n_env=1)
    np.random.seed(2)
    clip = _loader(cfg).sample_seq()  # the loader's sample dict (gender as a number, beta padded): what load_expert receives
    env = env_dict["humanoid_im"](cfg, init_expert=clip, data_specs=cfg.data_specs, mode="test")
    env.reset()
    b0 = env.prev_bquat.copy()
    rng = np.random.default_rng(0)
    for _ in range(3):
        env.step(rng.normal(scale=0.1, size=env.action_dim))
    d = env.data
    assert env.prev_bquat.shape == (96,) and not np.allclose(env.prev_bquat, b0)
    bq = env.get_body_quat()
    np.testing.assert_allclose(bq[:4], d.qpos[3:7])
    np.testing.assert_allclose(bq[4:8], quaternion_from_euler(d.qpos[7], d.qpos[8], d.qpos[9], "rzyx"))
    ee = env.get_ee_pos(None).reshape(5, 3)
    for k, n in enumerate(SMPL_EE_NAMES):
        np.testing.assert_allclose(ee[k], d.body_xpos[env.model.body_names.index(n)])
    assert np.linalg.norm(env.get_ee_pos("root").reshape(5, 3), axis=1).max() < 1.5
    np.testing.assert_allclose(env.get_com(), d.xipos[env.model.body_names.index("Pelvis")])
    env.fail_safe()
    t = env.cur_t
    np.testing.assert_allclose(env.data.qpos, env.expert["qpos"][min(t, env.expert["len"] - 1)], atol=1e-12)
    env.vec.close()


def test_agent_iteration_on_the_generated_model_class(tmp_path):
    """A reference config carries no `self_collision` / `rel_joint_lm` key: the env then runs the model class Robot(cfg.robot_cfg)
    generates (body-body collisions on, Chest / shoulder excludes, tight knee / ankle / toe ranges: uhc/envs/humanoid_im.py:52-64,
    uhc/smpllib/smpl_parser.py:327-328).  One full agent iteration (rollout through the dense-row kernels and the general tier, GAE,
    PPO update, checkpoint, evaluation) on it, and nothing exceeds the capacity of the last tier."""
    import torch
    from uhc_amd import sim as S
    from uhc_amd.agents import agent_dict
    torch.set_default_dtype(torch.float64)
    cfg = _cfg(tmp_path)
    cfg.robot_cfg = {"mesh": True, "model": "smpl"}  # config/release/uhc_implicit_shape.yml:92-98
    agent = agent_dict[cfg.agent_name](cfg, torch.float64, torch.device("cuda", 0), data_loader=_loader(cfg))
    m = agent.env.model
    assert (np.asarray(m.geom_contype)[1:] == 1).all() and m.nexclude == 2
    assert m.jnt_range[m.joint_names.index("R_Knee_z")] == pytest.approx([-np.pi / 16, np.pi / 16])
    info = agent.optimize_policy(0)
    log = info["log"]
    assert log.num_steps == 64 * 8 and 0.0 < log.avg_c_reward <= 1.0 and np.isfinite(log.avg_c_info).all()
    assert int(agent.env.sim.field(S.F_FAIL).sum().item()) == 0 and int(agent.env.sim.field(S.F_EFC_OVERFLOW).sum().item()) == 0
    assert "log_eval" in info and os.path.exists(os.path.join(cfg.model_dir, "iter_0001.p"))
    agent.env.close()


def test_agent_runs_every_clip_on_the_model_generated_from_its_beta(tmp_path):
    """The reference's load_expert rebuilds the MuJoCo model from the clip's beta / gender (reset_robot, uhc/envs/humanoid_im.py:154-190).
    Here the generator runs once per distinct (beta, gender) when the agent is built (body_provider: a synthetic stand-in for the
    licensed SMPL forward pass), the models share the batch, and the device switches an env's model blob when it starts a clip."""
    import torch
    from uhc_amd import sim as S
    from uhc_amd.agents import agent_dict
    from uhc_amd.data_loaders.synthetic import make_synthetic_body_provider
    torch.set_default_dtype(torch.float64)
    cfg = _cfg(tmp_path)
    cfg.robot_cfg = {"mesh": True, "model": "smpl"}
    dl = _loader(cfg)
    agent = agent_dict[cfg.agent_name](cfg, torch.float64, torch.device("cuda", 0), data_loader=dl, body_provider=make_synthetic_body_provider())
    env = agent.env
    assert len(env.models) == len(dl.data_keys) == 6  # six clips, six betas, six bodies
    masses = [float(m.body_mass.sum()) for m in env.models]
    assert max(masses) - min(masses) > 1.0
    assert all((np.asarray(m.geom_contype)[1:] == 1).all() for m in env.models)
    info = agent.optimize_policy(0)
    log = info["log"]
    assert log.num_steps == 64 * 8 and 0.0 < log.avg_c_reward <= 1.0
    assert int(env.sim.field(S.F_FAIL).sum().item()) == 0 and int(env.sim.field(S.F_EFC_OVERFLOW).sum().item()) == 0
    # each env sits on the blob of the clip it is running (uhc_env_set_clip_models): total mass on the root translation of qM
    qm00 = env.sim.field(S.F_QM)[:, 0].cpu().numpy()
    assert len(set(np.round(qm00, 6))) > 1 and all(min(abs(q - m) for m in masses) < 1e-9 for q in qm00)
    assert "log_eval" in info
    env.close()


def test_facade_load_expert_rebuilds_the_robot(tmp_path):
    """HumanoidEnv.load_expert(expert, reload_robot=True) -> reset_robot: a clip with another beta / gender gets the model generated from
    it (uhc/envs/humanoid_im.py:154-190, 182-190); the same beta keeps the model; reload_robot=False never rebuilds."""
    import torch
    from uhc_amd.data_loaders.synthetic import make_synthetic_body_provider
    from uhc_amd.envs import env_dict
    torch.set_default_dtype(torch.float64)
    cfg = _cfg(tmp_path)
    cfg.robot_cfg = {"mesh": True, "model": "smpl"}
    dl = _loader(cfg)
    keys = list(dl.data_keys)
    e0, e1 = dl.get_sample_from_key(keys[0], full_sample=True), dl.get_sample_from_key(keys[1], full_sample=True)
    env = env_dict["humanoid_im"](cfg, init_expert=e0, data_specs=cfg.data_specs, mode="test", body_provider=make_synthetic_body_provider())
    m0 = float(env.model.body_mass.sum())
    assert (np.asarray(env.model.geom_contype)[1:] == 1).all()
    env.reset()
    env.step(np.zeros(env.action_dim))
    env.load_expert(e1)  # another beta: another body
    m1 = float(env.model.body_mass.sum())
    assert abs(m1 - m0) > 1e-3 and env.vec.model is env.model
    obs = env.reset()
    assert np.isfinite(obs).all()
    np.testing.assert_allclose(env.data.qpos, env.expert["qpos"][0], atol=1e-12)
    vec = env.vec
    env.load_expert(e1)
    assert env.vec is vec  # same beta: nothing rebuilt
    env.load_expert(e0, reload_robot=False)
    assert env.vec is vec and float(env.model.body_mass.sum()) == m1
    env.vec.close()


def test_agent_iteration_on_the_ball_joint_humanoid(tmp_path):
    """config/copycat_ball/copycat_ball_1.yml's env: robot.ball, action_type torque (tq_mul 4), no residual force, no meta-PD, reward
    world_rfc_implicit_quat, obs_v 2 -> get_full_obs_v2_quat (534 numbers).  One agent iteration on it (self-collision on, as every
    generated model): the env layer that makes configs[4] a rollout and not a physics-only loop."""
    import torch
    from uhc_amd import sim as S
    from uhc_amd.agents import agent_dict
    torch.set_default_dtype(torch.float64)
    cfg = _cfg(tmp_path)
    cfg.robot_cfg = {"mesh": True, "model": "smpl", "ball": True}
    cfg.action_type, cfg.residual_force, cfg.meta_pd, cfg.meta_pd_joint = "torque", False, False, False
    cfg.reward_id, cfg.obs_v = "world_rfc_implicit_quat", 2
    cfg.cfg_dict["tq_mul"] = 4
    cfg.env_init_noise = 0.0
    # WITH the evaluation pass (checkpoint + eval_policy in the same iteration): eval_seqs' fail-safe teleports a failed clip to the expert
    # pose -- in round 3 it handed the 76-number hinge pose to the 99-number ball-joint model (a device read past the tensor: a GPU fault
    # and SIGABRT in nine fresh processes out of ten).  The pose now has the model's width, SimBatch.set_state refuses any other, and
    # tools/r04_pass.sh `loop` runs this test 20 x as the first test of a fresh process (profiles/r04_*_ball_agent_loop.txt)
    cfg.save_n_epochs = 1
    agent = agent_dict[cfg.agent_name](cfg, torch.float64, torch.device("cuda", 0), data_loader=_loader(cfg))
    env = agent.env
    assert env.use_quat and (env.model.nq, env.model.nv) == (99, 75) and agent.state_dim == 534 and agent.action_dim == 69
    info = agent.optimize_policy(0)
    log = info["log"]
    assert log.num_steps == 64 * 8 and 0.0 < log.avg_c_reward <= 1.0 and np.isfinite(log.avg_c_info).all()
    assert "log_eval" in info and np.isfinite(list(info["log_eval"][0].values())[0]["mpjpe"])
    ev = next(iter(agent._eval_envs.values()))  # the evaluation batch: its fail-safe teleports went through set_state at the model's width
    assert ev.use_quat and int(ev.sim.field(S.F_FAIL).sum().item()) == 0
    assert int(env.sim.field(S.F_FAIL).sum().item()) == 0 and int(env.sim.field(S.F_EFC_OVERFLOW).sum().item()) == 0
    q = env.sim.field(S.F_QPOS).cpu().numpy()
    assert np.abs(np.linalg.norm(q[:, 3:99].reshape(-1, 24, 4), axis=2) - 1).max() < 1e-9
    env.close()
