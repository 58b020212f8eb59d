"""CPU tests: this build's host code and the oracle against golden vectors produced by the imported
reference (tools/gen_golden.py; fixtures under tests/golden/)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def expert_from(f):
    e = {k[2:]: f[k] for k in f.files if k.startswith("f_")}
    e["len"] = int(e["len"])
    return e


REWARD_W = dict(w_p=0.3, w_v=0.1, w_e=0.45, w_c=0.1, w_vf=0.05, k_p=2.0, k_v=0.005, k_e=5.0, k_c=100.0, k_vf=1.0)


def test_quaternion_and_heading_helpers():
    from uhc_amd.utils import math_utils as mu, transformation as tf
    g = load("g1_math")
    q, p, v, eul = g["q"], g["p"], g["v"], g["eul"]

    def chk(name, val, tol=1e-13):
        np.testing.assert_allclose(np.asarray(val), g[name], atol=tol, rtol=0, err_msg=name)

    chk("quaternion_multiply", [tf.quaternion_multiply(a, b) for a, b in zip(q, p)])
    chk("quaternion_inverse", [tf.quaternion_inverse(a * 1.3) for a in q])
    chk("quaternion_matrix", [tf.quaternion_matrix(a)[:3, :3] for a in q])
    chk("quaternion_from_euler_rzyx", tf.quaternion_from_euler_rzyx(eul[:, 0], eul[:, 1], eul[:, 2]))
    chk("rotation_from_quaternion", [tf.rotation_from_quaternion(a) for a in q])
    chk("get_heading", [mu.get_heading(a) for a in q])
    chk("get_heading_q", [mu.get_heading_q(a) for a in q])
    chk("de_heading", [mu.de_heading(a) for a in q])
    chk("transform_vec_root", [mu.transform_vec(b, a, "root") for a, b in zip(q, v)])
    chk("transform_vec_heading", [mu.transform_vec(b, a, "heading") for a, b in zip(q, v)])
    chk("quat_mul_vec", [mu.quat_mul_vec(a, b) for a, b in zip(q, v)])
    out = mu.transform_vec_batch(g["vb"], q[0], "root")
    assert out.shape == (3, 24)  # component-major, SURVEY 3.5
    chk("transform_vec_batch_root", out)
    d = mu.multi_quat_diff(q[:24].ravel(), p[:24].ravel())
    chk("multi_quat_diff", d)
    chk("multi_quat_norm", mu.multi_quat_norm(d))
    chk("get_angvel_fd", mu.get_angvel_fd(p[:24].ravel(), q[:24].ravel(), 1 / 30), tol=1e-11)
    # doctest known answers of the reference's transformation.py
    chk("kat_about_axis", tf.quaternion_about_axis(0.123, [1, 0, 0]))
    chk("kat_multiply", tf.quaternion_multiply([4, 1, -2, 3], [8, -5, 6, 7]))
    np.testing.assert_allclose(g["kat_multiply"], [28, -44, -14, 48])
    np.testing.assert_allclose(g["kat_about_axis"], [0.99810947, 0.06146124, 0, 0], atol=1e-8)


def test_smpl_to_qpose_matches_reference_and_shipped_clip(model):
    from uhc_amd.smpllib.smpl_mujoco import smpl_to_qpose
    g = load("g2_smpl_to_qpose_standing")
    q = smpl_to_qpose(g["pose_aa"], model)
    np.testing.assert_allclose(q, g["qpos"], atol=2e-6)
    # implicit golden vector of the reference: standing_neutral.qpos == smpl_to_qpose(pose_aa[10])
    np.testing.assert_allclose(q[10, 3:], g["ref_qpos"][3:], atol=2e-6)
    g3 = load("g3_qpos_fk")
    np.testing.assert_allclose(smpl_to_qpose(g3["pose_aa"], model, trans=g3["trans"]), g3["f_qpos"], atol=2e-6)


def test_expert_features_qpos_fk(model):
    import torch
    from uhc_amd.smpllib.torch_smpl_humanoid import Humanoid
    g = load("g3_qpos_fk")
    f = Humanoid(model=model).qpos_fk(torch.from_numpy(g["f_qpos"]))
    assert f["len"] == int(g["f_len"])
    for k in ("qpos", "qvel", "wbpos", "wbquat", "bquat", "body_com", "rlinv", "rlinv_local", "rangv", "bangvel", "ee_wpos", "ee_pos",
              "com", "height_lb"):
        np.testing.assert_allclose(np.asarray(f[k]), g["f_" + k], atol=1e-11, rtol=0, err_msg=k)


def test_oracle_kinematics_matches_reference_fk(model):
    """The reference's own independent FK (torch_smpl_humanoid.py:303-362; pinned above through Humanoid.qpos_fk)
    pins the oracle's P1 stage.  The reference FK does not normalise the root quaternion (MuJoCo does), so the
    comparison is exact on unit quaternions and 2e-6 on the raw fixture."""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd.smpllib.torch_smpl_humanoid import Humanoid
    g = load("g3_qpos_fk")
    o = OracleSim(model)
    qn = g["f_qpos"].copy()
    qn[:, 3:7] /= np.linalg.norm(qn[:, 3:7], axis=1, keepdims=True)
    f = Humanoid(model=model).qpos_fk(torch.from_numpy(qn))
    for t in (0, 13, 39):
        o.set("qpos", qn[t])
        o.call("kinematics")
        np.testing.assert_allclose(o.get("xpos")[3:], f["wbpos"][t], atol=1e-12)
        np.testing.assert_allclose(o.get("xipos")[3:], f["body_com"][t], atol=1e-12)
        wq, rq = o.get("xquat")[4:].reshape(-1, 4), f["wbquat"][t].reshape(-1, 4)
        np.testing.assert_allclose(np.abs((wq * rq).sum(1)), 1.0, atol=1e-12)  # same rotation (q ~ -q)
        o.set("qpos", g["f_qpos"][t])
        o.call("kinematics")
        np.testing.assert_allclose(o.get("xpos")[3:], g["f_wbpos"][t], atol=2e-6)


def test_env_oracle_observation_reward_termination(model):
    from oracle import env_oracle as E
    from uhc_amd.smpllib.smpl_mujoco import SMPLConverter
    g, f = load("g4_g6_obs_reward"), load("g3_qpos_fk")
    expert = expert_from(f)
    jw = SMPLConverter(model, model).get_new_diff_weight()
    for c in range(int(g["ncase"])):
        p = f"c{c}_"
        t = int(g[p + "cur_t"])
        obs = E.full_obs_v2(g[p + "qpos"], g[p + "qvel"], g[p + "xpos"], g[p + "xquat"], expert, t, 0, g[p + "beta"], float(g["gender"]))
        assert obs.shape == (657,)
        np.testing.assert_allclose(obs, g[p + "obs"], atol=1e-13)
        np.testing.assert_allclose(E.get_body_quat(g[p + "qpos"]), g[p + "bquat"], atol=1e-14)
        bd = E.calc_body_diff(g[p + "xpos"], expert["wbpos"][E.expert_index(t, 0, expert["len"])], jw)
        assert bd == pytest.approx(float(g[p + "body_diff"]), abs=1e-14)
        r, parts = E.world_rfc_implicit_reward(g[p + "qpos"], g[p + "xpos"], g[p + "xipos"], g[p + "prev_bquat"], g[p + "action"], expert,
                                               t, 0, model.timestep * 15, jw[1:], REWARD_W)
        assert r == pytest.approx(float(g[p + "reward"]), abs=1e-13)
        np.testing.assert_allclose(parts, g[p + "reward_info"], atol=1e-12)


def test_env_oracle_obs_v1_v6_and_explicit_reward(model):
    from oracle import env_oracle as E
    from uhc_amd.smpllib.smpl_mujoco import SMPLConverter
    g, f = load("g4b_obs_variants"), load("g3_qpos_fk")
    expert = expert_from(f)
    jw = SMPLConverter(model, model).get_new_diff_weight()
    for c in range(int(g["ncase"])):
        p = f"c{c}_"
        t = int(g[p + "cur_t"])
        o1 = E.full_obs_v1(g[p + "qpos"], g[p + "qvel"], g[p + "xpos"], g[p + "xquat"], g[p + "xipos"], expert, t, 0)
        assert o1.shape == (784,)
        np.testing.assert_allclose(o1, g[p + "obs_v1"], atol=1e-13)
        o6 = E.full_obs_v6(g[p + "qpos"], g[p + "qvel"], g[p + "xpos"], expert, t, 0, g[p + "beta"], float(g["gender"]))
        assert o6.shape == (401,)
        np.testing.assert_allclose(o6, g[p + "obs_v6"], atol=1e-13)
        o3 = E.full_obs_v3(g[p + "qpos"], g[p + "qvel"], g[p + "xpos"], g[p + "xquat"], expert, t, 0, g[p + "beta"], float(g["gender"]), fut_frames=3, skip=4)
        assert o3.shape == (3 * 657,)
        np.testing.assert_allclose(o3, g[p + "obs_v3"], atol=1e-13)
        r, parts = E.world_rfc_explicit_reward(g[p + "qpos"], g[p + "xpos"], g[p + "xipos"], g[p + "prev_bquat"], g[p + "action"], expert,
                                               t, 0, model.timestep * 15, jw[1:], REWARD_W)
        assert r == pytest.approx(float(g[p + "reward_explicit"]), abs=1e-13)
        np.testing.assert_allclose(parts, g[p + "reward_explicit_info"], atol=1e-12)


def test_oracle_pd_controller_and_rfc(model, ctrl):
    """compute_torque / compute_desired_accel / rfc_implicit of the reference vs the C oracle."""
    from oracle.physics import OracleSim
    g = load("g5_pd_rfc")
    o = OracleSim(model, ctrl)
    o.set("qpos", g["qpos"])
    o.set("qvel", g["qvel"])
    M, qM = g["M"], np.zeros(model.nM)
    for i in range(model.nv):
        adr, j = model.dof_madr[i], i
        while j >= 0:
            qM[adr] = M[i, j]
            adr += 1
            j = model.dof_parentid[j]
    o.set("qM", qM)
    o.set("qfrc_bias", g["C"])
    for it in (0, 7, 14):
        ref = np.clip(g[f"torque_{it}"], -g["torque_lim"], g["torque_lim"])
        np.testing.assert_allclose(o.pd_torque(g["action"], g["target_base"], it), ref, atol=1e-9)
    np.testing.assert_allclose(o.rfc_implicit(g["action"]), g["qfrc_applied"], atol=1e-12)
    np.testing.assert_allclose(np.array([ctrl.jkp[i] for i in range(69)]), g["jkp"])
    np.testing.assert_allclose(np.array([ctrl.torque_lim[i] for i in range(69)]), g["torque_lim"])


def test_eval_metrics_match_reference():
    from uhc_amd.smpllib.smpl_eval import compute_metrics
    g = load("g11_metrics")
    out = compute_metrics(dict(pred=g["pred"], gt=g["gt"], pred_jpos=g["pred_jpos"], gt_jpos=g["gt_jpos"], fail_safe=False, percent=1))
    for k, v in out.items():
        np.testing.assert_allclose(np.asarray(v, dtype=float), g["m_" + k], atol=1e-9, err_msg=k)
    assert bool(out["succ"][0]) is True


# ---- G9: clip sampling (dataset_amass_single.py:27-317) ---------------------------------------------------
def _g9_loader():
    from uhc_amd.data_loaders.dataset_amass_single import DatasetAMASSSingle
    g = load("g9_dataset")
    genders = ["neutral", "male", "female"]
    pk = {}
    for i in range(8):
        pk[f"0-SYN_{i:02d}_poses"] = dict(pose_aa=g[f"in{i}_pose_aa"], pose_6d=g[f"in{i}_pose_6d"], trans=g[f"in{i}_trans"], beta=g[f"in{i}_beta"],
                                          gender=genders[i % 3], seq_name=f"SYN_{i:02d}")
    specs = dict(file_path="syn.pkl", test_file_path="syn.pkl", t_min=15, t_max=30, mode="all")
    return g, list(pk.keys()), DatasetAMASSSingle(specs, "train", pickle_data=pk)


def test_dataset_sampling_sequences_match_reference():
    import random
    g, keys, dl = _g9_loader()
    assert [keys.index(k) for k in dl.data_keys] == g["data_keys"].tolist()
    assert [keys.index(k) for k, _ in dl.sample_keys] == g["sample_keys"].tolist()
    random.seed(3); np.random.seed(3)
    rows = []
    for _ in range(24):
        s = dl.sample_seq(freq_dict=None)
        rows.append([keys.index(dl.curr_key), dl.fr_start, dl.fr_end, s["pose_aa"].shape[0], s["beta"].shape[1], s["gender"][0], int(s["has_obj"]), s["num_obj"]])
    np.testing.assert_array_equal(np.array(rows), g["uniform"])
    np.testing.assert_array_equal(s["pose_aa"], g["uniform_first_pose"])
    fd = {k: [] for k in dl.data_keys}
    for ki, p, s0 in g["fd_flat"]:
        fd[dl.data_keys[int(ki)]].append([float(p), int(s0)])
    for name, prec in (("weighted", False), ("precision", True)):
        random.seed(5); np.random.seed(5)
        rows = []
        for _ in range(24):
            dl.sample_seq(freq_dict=fd, sampling_temp=0.2, sampling_freq=0.5, precision_mode=prec)
            rows.append([keys.index(dl.curr_key), dl.fr_start, dl.fr_end])
        np.testing.assert_array_equal(np.array(rows), g[name], err_msg=name)
    s = dl.get_sample_from_key(dl.data_keys[2], full_sample=True)
    np.testing.assert_array_equal([dl.fr_start, dl.fr_end, s["pose_aa"].shape[0]], g["full"])
    s = dl.get_sample_from_key(dl.data_keys[0], fr_start=7)
    np.testing.assert_array_equal([dl.fr_start, dl.fr_end, s["pose_aa"].shape[0]], g["fixed_start"])
    np.testing.assert_array_equal([keys.index(dl.iter_seq()["seq_name"]) for _ in range(8)], g["iter"])


# ---- G10: configuration parsing (copycat_config.py:12-168, base_config.py:9-62) ----------------------------
def test_config_attributes_match_reference(tmp_path):
    import json
    from uhc_amd.utils.config_utils.copycat_config import Config
    g = load("g10_config")
    ids = sorted({k.split("__")[0] for k in g.files})
    assert len(ids) == 7
    for cid in ids:
        cfg = Config(cfg_id=cid, base_dir=str(tmp_path), cfg_dict=json.loads(str(g[f"{cid}__yml"])))
        want = json.loads(str(g[f"{cid}__scalars"]))
        assert want.pop("adv_clip_is_inf") == bool(np.isinf(cfg.adv_clip))
        for k, v in want.items():
            assert getattr(cfg, k) == v, (cid, k)
        for k in g.files:
            if k.startswith(cid + "__") and k.split("__")[1] not in ("yml", "scalars", "adaptive"):
                np.testing.assert_allclose(np.asarray(getattr(cfg, k.split("__")[1]), dtype=float), g[k], rtol=0, atol=0, err_msg=k)
        for it, nr, ls, lr in g[f"{cid}__adaptive"]:
            cfg.update_adaptive_params(int(it))
            np.testing.assert_allclose([cfg.adp_noise_rate, cfg.adp_log_std, cfg.adp_policy_lr], [nr, ls, lr], rtol=1e-15, atol=0)


def test_env_oracle_obs_v4():
    """G4d: get_full_obs_v4 of the imported reference (uhc/envs/humanoid_im.py:769-861) -- the full vector, the (23, 26) local block and the global block."""
    from oracle import env_oracle as E
    g, f = load("g4d_obs_v4"), load("g3_qpos_fk")
    expert = {k[2:]: f[k] for k in f.files if k.startswith("f_")}
    expert["len"] = expert["qpos"].shape[0]
    for c in range(int(g["ncase"])):
        p, t = f"c{c}_", int(g[f"c{c}_cur_t"])
        full, local, glob = E.full_obs_v4(g[p + "qpos"], g[p + "qvel"], g[p + "xpos"], g[p + "xquat"], expert, t, 0, g[p + "beta"], float(g["gender"]))
        assert full.shape == (643,) and local.shape == (23, 26) and glob.shape == (45,)
        np.testing.assert_allclose(full, g[p + "obs_full"], atol=1e-13)
        np.testing.assert_allclose(local, g[p + "local_obs"], atol=1e-13)
        np.testing.assert_allclose(glob, g[p + "global_obs"], atol=1e-13)


def test_env_oracle_obs_v0_v5_and_remaining_rewards(model):
    """G4c: get_full_obs (v0), get_full_obs_v5 and the reward ids implicit_quat, v1_mul, explicit_mul, v2, v3 of the imported reference."""
    from oracle import env_oracle as E
    from uhc_amd.smpllib.smpl_mujoco import SMPLConverter
    g, f = load("g4c_more_variants"), load("g3_qpos_fk")
    expert = {k[2:]: f[k] for k in f.files if k.startswith("f_")}
    expert["len"] = expert["qpos"].shape[0]
    conv = SMPLConverter(model, model)
    w = dict(w_p=0.3, w_v=0.1, w_e=0.45, w_c=0.1, w_vf=0.05, k_p=2.0, k_v=0.005, k_e=5.0, k_c=100.0, k_vf=1.0)
    w23 = {str(k): float(v) for k, v in zip(g["w_v23_keys"], g["w_v23_vals"])}
    dt = model.timestep * 15
    for c in range(int(g["ncase"])):
        p, t = f"c{c}_", int(g[f"c{c}_cur_t"])
        o0 = E.full_obs_v0(g[p + "qpos"], g[p + "qvel"], expert, t, 0, obs_heading=True)
        np.testing.assert_allclose(o0, g[p + "obs_v0"], atol=1e-13)
        o5 = E.full_obs_v5(g[p + "qpos"], g[p + "qvel"], g[p + "xpos"], g[p + "xquat"], expert, t, 0, g[p + "beta"], float(g["gender"]))
        assert o5.shape == g[p + "obs_v5"].shape
        np.testing.assert_allclose(o5, g[p + "obs_v5"], atol=1e-13)
        args = (g[p + "qpos"], g[p + "xpos"], g[p + "xipos"], g[p + "prev_bquat"], g[p + "action"], expert, t, 0, dt, conv.get_new_diff_weight()[1:], w)
        r, parts = E.world_rfc_implicit_reward(*args)  # the _quat id is the same function body
        np.testing.assert_allclose([r, *parts], [g[p + "world_rfc_implicit_quat"], *g[p + "world_rfc_implicit_quat_info"]], atol=1e-13)
        r, parts = E.world_rfc_mul_reward(False, *args)
        np.testing.assert_allclose([r, *parts], [g[p + "world_rfc_implicit_v1_mul"], *g[p + "world_rfc_implicit_v1_mul_info"]], atol=1e-13)
        args_e = args[:4] + (g[p + "action_explicit"],) + args[5:]
        r, parts = E.world_rfc_mul_reward(True, *args_e)
        np.testing.assert_allclose([r, *parts], [g[p + "world_rfc_explicit_mul"], *g[p + "world_rfc_explicit_mul_info"]], atol=1e-13)
        for v3, name in ((False, "world_rfc_implicit_v2"), (True, "world_rfc_implicit_v3")):
            r, parts = E.world_rfc_implicit_v2_v3(v3, g[p + "qpos"], g[p + "xpos"], g[p + "xquat"], g[p + "xipos"], g[p + "prev_bquat"], g[p + "action"],
                                                  expert, t, 0, dt, w23, g["w_v23_jpos_diffw"])
            np.testing.assert_allclose([r, *parts], [g[p + name], *g[p + name + "_info"]], atol=1e-13)


def test_process_amass_db_matches_reference():
    """G13: uhc_amd/data_process/process_amass_db.py against the reference's process_qpos_list + split rule on a synthetic AMASS
    database (frame-rate stride, occlusion truncation / rejection, minimum length, float32 6-D rotations, 'vald' slip)."""
    from uhc_amd.data_process.process_amass_db import process_qpos_list, split_amass
    g = load("g13_process_amass")
    names = [str(n) for n in g["names"]]
    db = [(n, {"poses": g[f"db_{n}_poses"], "trans": g[f"db_{n}_trans"], "betas": g[f"db_{n}_betas"], "gender": str(g[f"db_{n}_gender"]),
               "mocap_framerate": float(g[f"db_{n}_fr"])}) for n in names]
    occ = {}
    for k, issue, i0 in zip(g["occ_keys"], g["occ_issue"], g["occ_idx0"]):
        occ[str(k)] = {"issue": str(issue)}
        if int(i0) >= 0:
            occ[str(k)]["idxes"] = [int(i0), int(i0) + 1]
    res = process_qpos_list(db, occ, log=lambda *a: None)
    assert list(res) == [str(k) for k in g["kept"]]
    for k, v in res.items():
        np.testing.assert_array_equal(v["pose_aa"], g[f"res_{k}_pose_aa"])
        np.testing.assert_array_equal(v["trans"], g[f"res_{k}_trans"])
        np.testing.assert_array_equal(v["beta"], g[f"res_{k}_beta"])
        assert v["pose_6d"].dtype == np.float32 and v["pose_6d"].shape == g[f"res_{k}_pose_6d"].shape
        np.testing.assert_allclose(v["pose_6d"], g[f"res_{k}_pose_6d"], atol=5e-7)
        assert v["seq_name"] == k and v["height_fixed"] is False
    train, test, valid = split_amass(res, log=lambda *a: None)
    want = dict(zip([str(k) for k in g["split_keys"]], [str(v) for v in g["split_vals"]]))
    got = {**{k: "train" for k in train}, **{k: "test" for k in test}, **{k: "valid" for k in valid}}
    assert got == want and valid == {}   # HumanEva (a validation set) lands in train: the reference's 'vald' / 'valid' slip


def test_ball_env_quaternion_paths_match_reference(model):
    """G14: the ball-joint env (robot.ball / use_quat, config/copycat_ball): expert conversion with use_quat, get_body_quat on the ball
    qpos, world_rfc_implicit_quat, and get_full_obs_v2_quat given the quaternion expert pose (the pose the function was written for;
    the reference hands it the Euler one and raises -- tools/gen_golden.py: g14_ball_env)."""
    from oracle import env_oracle as E
    from uhc_amd.model.mjcf import ball_variant
    from uhc_amd.smpllib.smpl_mujoco import SMPLConverter, smpl_to_qpose
    g = load("g14_ball_env")
    ball = ball_variant(model)
    # (the reference converts axis-angle -> rotation with a float32-grade routine: 2e-6, as for the Euler pose of G2)
    np.testing.assert_allclose(smpl_to_qpose(g["pose_aa"], ball, trans=g["trans"], count_offset=True, use_quat=True), g["qpos_quat"], atol=2e-6)
    expert = {k[2:]: g[k] for k in g.files if k.startswith("f_")}
    expert["len"] = int(expert["len"])
    expert["qpos_quat"] = g["qpos_quat"]
    # the joints of the quaternion expert pose ARE the expert's local body quaternions (bquat), sign included: the bank record needs no
    # new field for them; the ROOT quaternions of the two poses come out of different routines of the reference (1.5e-3 rad apart)
    np.testing.assert_allclose(g["qpos_quat"][:, 7:], expert["bquat"][:, 4:], atol=1e-12)
    root_dot = np.abs((g["qpos_quat"][:, 3:7] * expert["bquat"][:, :4]).sum(1))
    assert root_dot.min() > 1 - 1e-6 and root_dot.min() < 1 - 1e-9
    # the fail-safe teleport of the evaluation loop puts an env on the expert pose in the MODEL's coordinates: for the ball-joint model
    # that is the quaternion pose (99 numbers) rebuilt from the frame record, not the record's 76 hinge angles
    from uhc_amd import sim as S
    q_rec = np.array(expert["qpos"], dtype=np.float64, copy=True)
    q_rec[:, 3:7] = g["qpos_quat"][:, 3:7]
    fr = S.pack_expert_frames(dict(expert, qpos=q_rec))
    qp, qv = S.expert_pose_of_frames(fr, True)
    assert qp.shape == (expert["len"], 99) and qv.shape == (expert["len"], 75)
    np.testing.assert_allclose(qp, g["qpos_quat"], atol=1e-12)
    qh, _ = S.expert_pose_of_frames(fr, False)
    np.testing.assert_array_equal(qh, q_rec)
    jw = SMPLConverter(model, model).get_new_diff_weight()
    w = dict(w_p=0.3, w_v=0.1, w_e=0.45, w_c=0.1, w_vf=0.05, k_p=2.0, k_v=0.005, k_e=5.0, k_c=100.0, k_vf=1.0)
    for c in range(int(g["ncase"])):
        p = f"c{c}_"
        t = int(g[p + "cur_t"])
        obs = E.full_obs_v2_quat(g[p + "qpos"], g[p + "qvel"], g[p + "xpos"], g[p + "xquat"], expert, t, 0, g[p + "beta"], float(g["gender"]))
        assert obs.shape == (534,)
        np.testing.assert_allclose(obs, g[p + "obs"], atol=1e-12)
        np.testing.assert_allclose(E.get_body_quat_ball(g[p + "qpos"]), g[p + "bquat"], atol=0)
        r, parts = E.world_rfc_implicit_reward(g[p + "qpos"], g[p + "xpos"], g[p + "xipos"], g[p + "prev_bquat"], g[p + "action"], expert, t, 0,
                                               model.timestep * 15, jw[1:], w, vf_dim=0, ball=True)
        assert r == pytest.approx(float(g[p + "reward"]), abs=1e-12)
        np.testing.assert_allclose(parts, g[p + "reward_info"], atol=1e-12)
