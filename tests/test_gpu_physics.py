"""GPU parity: the HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


import os


@pytest.fixture(params=["fast", "general"], autouse=True)
def kernel_path(request):
    """Every parity test runs on the fast kernel (with automatic redo) and on the general kernel alone."""
    old = os.environ.get("UHC_FORCE_GENERAL")
    os.environ["UHC_FORCE_GENERAL"] = "1" if request.param == "general" else "0"
    yield request.param
    if old is None:
        os.environ.pop("UHC_FORCE_GENERAL", None)
    else:
        os.environ["UHC_FORCE_GENERAL"] = old


def _sim(model, ctrl, n):
    from uhc_amd.sim import SimBatch
    return SimBatch(model, ctrl, n)


def _states(standing, model, n, seed, noise=0.1, vel=0.5, lift=0.0):
    rng = np.random.default_rng(seed)
    qpos = np.tile(standing["qpos"], (n, 1))
    qpos[:, 7:] += rng.normal(scale=noise, size=(n, model.nu))
    qpos[:, 2] += lift
    qvel = rng.normal(scale=vel, size=(n, model.nv))
    return qpos, qvel


def test_forward_fields_match_oracle(model, ctrl, standing):
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    n = 16
    qpos, qvel = _states(standing, model, n, 1)
    b = _sim(model, ctrl, n)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    b.sync()
    g = {k: b.field(f).cpu().numpy() for k, f in dict(xpos=S.F_XPOS, xquat=S.F_XQUAT, xipos=S.F_XIPOS, qM=S.F_QM,
                                                     bias=S.F_QFRC_BIAS, qacc=S.F_QACC, ncon=S.F_NCON, nefc=S.F_NEFC).items()}
    for e in range(n):
        o = OracleSim(model, ctrl)
        o.set_state(qpos[e], qvel[e])
        assert g["ncon"][e] == o.geti("ncon") and g["nefc"][e] == o.geti("nefc")
        np.testing.assert_allclose(g["xpos"][e], o.get("xpos"), atol=1e-13)
        np.testing.assert_allclose(g["xquat"][e], o.get("xquat"), atol=1e-13)
        np.testing.assert_allclose(g["xipos"][e], o.get("xipos"), atol=1e-13)
        np.testing.assert_allclose(g["qM"][e], o.get("qM"), atol=1e-11)
        np.testing.assert_allclose(g["bias"][e], o.get("qfrc_bias"), atol=1e-9)
        # constrained acceleration: PGS is run to the same tolerance on both sides
        np.testing.assert_allclose(g["qacc"][e], o.get("qacc"), atol=1e-5, rtol=1e-6)


def test_free_flight_trajectory(model, ctrl, standing):
    """No contacts (lifted 100 m, falls ~14 m): smooth dynamics + PD + RFC, 50 env-steps, tight tolerance."""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    n = 4
    qpos, qvel = _states(standing, model, n, 2, lift=100.0)
    rng = np.random.default_rng(3)
    act = rng.normal(scale=0.2, size=(n, ctrl.action_dim))
    b = _sim(model, ctrl, n)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    a, tb = torch.from_numpy(act).cuda(), torch.from_numpy(qpos[:, 7:].copy()).cuda()
    os_ = [OracleSim(model, ctrl) for _ in range(n)]
    for e in range(n):
        os_[e].set_state(qpos[e], qvel[e])
    for t in range(50):
        b.simulate(a, tb)
        for e in range(n):
            os_[e].do_simulation(act[e], qpos[e, 7:])
    b.sync()
    gq, gv = b.field(S.F_QPOS).cpu().numpy(), b.field(S.F_QVEL).cpu().numpy()
    for e in range(n):
        np.testing.assert_allclose(gq[e], os_[e].get("qpos"), atol=1e-8)
        np.testing.assert_allclose(gv[e], os_[e].get("qvel"), atol=1e-7)


@pytest.mark.parametrize("sweep_cap", [100, 300, "exact"])
def test_contact_trajectory_200_steps(model, ctrl, standing, sweep_cap, kernel_path):
    """north_star bar: per-step qpos/qvel within 1e-4 of the CPU path over 200 env-steps, same seed.  With MuJoCo's default cap of
    100 sweeps both sides run the same number of sweeps; at 300 (the package default: converged) the
    tolerance test decides, and the two implementations may stop a sweep apart."""
    import dataclasses
    import torch
    if sweep_cap == "exact":  # solver 1: the QP's exact optimum on both sides (fast kernel: active set in registers; general: working sets)
        model = dataclasses.replace(model, solver=1)
    else:
        model = dataclasses.replace(model, iterations=sweep_cap)
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    n = 4
    qpos, qvel = _states(standing, model, n, 4, noise=0.02, vel=0.05)
    rng = np.random.default_rng(5)
    b = _sim(model, ctrl, n)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
    os_ = [OracleSim(model, ctrl) for _ in range(n)]
    for e in range(n):
        os_[e].set_state(qpos[e], qvel[e])
    worst_q = worst_v = 0.0
    redone = 0
    for t in range(200):
        act = rng.normal(scale=0.05, size=(n, ctrl.action_dim))
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        gq, gv = b.field(S.F_QPOS).cpu().numpy(), b.field(S.F_QVEL).cpu().numpy()
        redo = b.field(S.F_REDO).cpu().numpy() if sweep_cap == "exact" else np.zeros(n, dtype=int)
        for e in range(n):
            # solver 1 covers the fast kernel; in a step beyond its capacity the general kernel's working sets may give up in some
            # substeps and sweep (UHC_F_REDO bits 8+): the oracle takes solver 0 in exactly those
            redone += int(redo[e] != 0)
            os_[e].do_simulation(act[e], qpos[e, 7:], redo=redo[e])
            worst_q = max(worst_q, np.abs(gq[e] - os_[e].get("qpos")).max())
            worst_v = max(worst_v, np.abs(gv[e] - os_[e].get("qvel")).max())
    print(f"200-step parity: max|dqpos|={worst_q:.3e} max|dqvel|={worst_v:.3e} (env-steps through the general kernel: {redone})")
    assert worst_q < 1e-4 and worst_v < 1e-4


def test_exact_solver_forward(model, ctrl, standing, kernel_path):
    """solver 1 (active-set / block principal pivoting): the constrained acceleration equals the oracle's exact optimum far below
    the sweep tolerance, within a handful of factorisations, and differs from the 100-sweep PGS answer."""
    import dataclasses
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    mx = dataclasses.replace(model, solver=1)
    n = 32
    qpos, qvel = _states(standing, model, n, 21)
    b = _sim(mx, ctrl, n)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    b.sync()
    gacc, git, gn = b.field(S.F_QACC).cpu().numpy(), b.field(S.F_SOLVER_ITER).cpu().numpy(), b.field(S.F_NEFC).cpu().numpy()
    moved = 0
    for e in range(n):
        o = OracleSim(mx, ctrl)
        o.set_state(qpos[e], qvel[e])
        assert gn[e] == o.geti("nefc")
        np.testing.assert_allclose(gacc[e], o.get("qacc"), atol=1e-8, rtol=1e-9)
        if gn[e] and kernel_path == "fast":  # the same pivoting sequence as the oracle; the general kernel's working sets count differently
            assert 1 <= git[e] <= 12 and abs(int(git[e]) - o.geti("solver_iter")) <= 1
        p = OracleSim(model, ctrl)
        p.set_state(qpos[e], qvel[e])
        moved += np.abs(p.get("qacc") - o.get("qacc")).max() > 1e-6
    assert moved > 0  # the tolerance-terminated sweeps stop measurably short of the optimum
    b.close()


def test_explicit_rfc_trajectory(model, standing):
    """Explicit residual forces (one body-frame wrench per body, applied through the point Jacobian of the previous
    forward pass): applied generalized force and a 40-step trajectory against the oracle."""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    ctrl = S.make_ctrl(model, residual_force_mode="explicit")
    assert ctrl.rfc_mode == 2 and ctrl.action_dim == 69 + 24 * 9 + 30
    n = 4
    qpos, qvel = _states(standing, model, n, 14, noise=0.05, vel=0.2)
    rng = np.random.default_rng(15)
    b = _sim(model, ctrl, n)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
    os_ = [OracleSim(model, ctrl) for _ in range(n)]
    for e in range(n):
        os_[e].set_state(qpos[e], qvel[e])
    worst_q = worst_v = worst_f = 0.0
    for t in range(40):
        act = rng.normal(scale=0.05, size=(n, ctrl.action_dim))
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        gq, gv, gf = (b.field(f).cpu().numpy() for f in (S.F_QPOS, S.F_QVEL, S.F_QFRC_APPLIED))
        for e in range(n):
            os_[e].do_simulation(act[e], qpos[e, 7:])
            worst_q = max(worst_q, np.abs(gq[e] - os_[e].get("qpos")).max())
            worst_v = max(worst_v, np.abs(gv[e] - os_[e].get("qvel")).max())
            worst_f = max(worst_f, np.abs(gf[e] - os_[e].get("qfrc_applied")).max())
    print(f"explicit RFC 40-step parity: max|dqpos|={worst_q:.3e} max|dqvel|={worst_v:.3e} max|dqfrc_applied|={worst_f:.3e}")
    assert np.abs(gf).max() > 1.0  # the wrenches are really applied
    assert worst_q < 1e-8 and worst_v < 1e-7 and worst_f < 1e-8


def test_inactive_envs_untouched(model, ctrl, standing):
    import torch
    from uhc_amd import sim as S
    n = 8
    qpos, qvel = _states(standing, model, n, 6)
    b = _sim(model, ctrl, n)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    act = torch.zeros(n, ctrl.action_dim, dtype=torch.float64, device="cuda")
    tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
    active = torch.tensor([1, 0] * 4, dtype=torch.int32, device="cuda")
    before = b.field(S.F_QPOS).clone()
    b.simulate(act, tb, active)
    b.sync()
    after = b.field(S.F_QPOS)
    assert torch.equal(before[1::2], after[1::2])
    assert not torch.equal(before[0::2], after[0::2])


def test_many_contacts_take_the_redo_path(model, ctrl, standing, kernel_path):
    """A humanoid lying on the floor has more than 64 constraint rows: the fast kernel must hand the env
    to the general kernel and the result must still match the oracle."""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    n = 3
    qpos, qvel = _states(standing, model, n, 7, noise=0.02, vel=0.0)
    # rotate the root so the body lies face-down just above the floor (world x-axis rotation by -90 deg composed on the left)
    c, s_ = np.cos(-np.pi / 4), np.sin(-np.pi / 4)
    for e in range(n):
        w, x, y, z = qpos[e, 3:7]
        qpos[e, 3:7] = [c * w - s_ * x, c * x + s_ * w, c * y - s_ * z, c * z + s_ * y]
        qpos[e, 2] = 0.19 + 0.01 * e
    b = _sim(model, ctrl, n)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    act = np.zeros((n, ctrl.action_dim))
    tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
    os_ = [OracleSim(model, ctrl) for _ in range(n)]
    for e in range(n):
        os_[e].set_state(qpos[e], qvel[e])
    max_nefc = 0
    for t in range(5):
        b.simulate(torch.from_numpy(act).cuda(), tb)
        for e in range(n):
            os_[e].do_simulation(act[e], qpos[e, 7:])
    b.sync()
    gq = b.field(S.F_QPOS).cpu().numpy()
    gn = b.field(S.F_NEFC).cpu().numpy()
    checked = 0
    for e in range(n):
        max_nefc = max(max_nefc, os_[e].geti("max_nefc"))
        if os_[e].geti("max_nefc") <= 128 and os_[e].geti("max_ncon") <= 40:  # inside the general kernel's capacity
            assert gn[e] == os_[e].geti("nefc")
            np.testing.assert_allclose(gq[e], os_[e].get("qpos"), atol=1e-7)
            checked += 1
    assert max_nefc > 64 and checked > 0, f"scenario must exceed the fast kernel's capacity and stay inside the general one (max nefc={max_nefc}, checked={checked})"


def test_truncate_mode_keeps_heavy_contact_envs_in_the_fast_kernel(model, ctrl, standing, kernel_path):
    """Opt-in overflow mode: a lying humanoid (beyond the fast kernel's contact / row storage) keeps what fits, is flagged (in the launch where it happens), and no env
    waits for a second pass; envs inside the capacity are bit-identical in both modes."""
    import torch
    from uhc_amd import sim as S
    if kernel_path == "general":
        pytest.skip("the mode only affects the fast kernel")
    n = 4
    qpos, qvel = _states(standing, model, n, 7, noise=0.02, vel=0.0)
    c, s_ = np.cos(-np.pi / 4), np.sin(-np.pi / 4)
    for e in range(2):  # envs 0, 1 lie face-down just above the floor; envs 2, 3 stand
        w, x, y, z = qpos[e, 3:7]
        qpos[e, 3:7] = [c * w - s_ * x, c * x + s_ * w, c * y - s_ * z, c * z + s_ * y]
        qpos[e, 2] = 0.19 + 0.01 * e
    out = {}
    for trunc in (False, True):
        b = _sim(model, ctrl, n)
        b.set_overflow_mode(trunc)
        b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
        act = torch.zeros(n, ctrl.action_dim, dtype=torch.float64, device="cuda")
        tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
        for _ in range(5):
            b.simulate(act, tb)
        b.sync()
        out[trunc] = {k: b.field(f).cpu().numpy().copy() for k, f in dict(q=S.F_QPOS, ncon=S.F_NCON, nefc=S.F_NEFC, ov=S.F_EFC_OVERFLOW, fail=S.F_FAIL).items()}
        b.close()
    ex, tr = out[False], out[True]
    assert not ex["ov"].any()                                     # exact mode: the general kernel took the heavy envs, nothing dropped
    assert tr["ncon"].max() <= 16 and tr["nefc"].max() <= 64      # truncate mode: everything stayed inside the fast kernel
    assert tr["ov"][:2].any() and not tr["ov"][2:].any() and not tr["fail"].any() and np.isfinite(tr["q"]).all()
    np.testing.assert_array_equal(tr["q"][2:], ex["q"][2:])                                    # envs inside the capacity: identical
    assert np.abs(tr["q"][:2, 2] - ex["q"][:2, 2]).max() < 0.05                                # the truncated bodies still rest on the floor


def test_per_env_models_share_one_batch(model, ctrl, standing, kernel_path):
    """smpl_shape-style batch: envs with differently scaled bodies (mass ~ s^3, inertia ~ s^5) in one launch."""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import scale_model
    scales = [1.0, 0.9, 1.12]
    models = [model] + [scale_model(model, s) for s in scales[1:]]
    env_model = [0, 1, 2, 1, 0, 2]
    n = len(env_model)
    qpos, qvel = _states(standing, model, n, 9, noise=0.03, vel=0.1)
    for e in range(n):
        qpos[e, :3] *= scales[env_model[e]]
    rng = np.random.default_rng(10)
    act = rng.normal(scale=0.1, size=(n, ctrl.action_dim))
    b = S.SimBatch(models, ctrl, n, env_model=env_model)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
    os_ = [OracleSim(models[env_model[e]], ctrl) for e in range(n)]
    for e in range(n):
        os_[e].set_state(qpos[e], qvel[e])
    for t in range(10):
        b.simulate(torch.from_numpy(act).cuda(), tb)
        for e in range(n):
            os_[e].do_simulation(act[e], qpos[e, 7:])
    b.sync()
    gq, gm = b.field(S.F_QPOS).cpu().numpy(), b.field(S.F_QM).cpu().numpy()
    for e in range(n):
        np.testing.assert_allclose(gq[e], os_[e].get("qpos"), atol=1e-9)
        np.testing.assert_allclose(gm[e], os_[e].get("qM"), atol=1e-9)
    assert abs(gm[1, 0] / gm[0, 0] - 0.9 ** 3) < 1e-9  # total mass on the root translation scales with s^3


def test_per_body_scaled_models_share_one_batch(model, ctrl, standing, kernel_path):
    """configs[3] as `bench.py --shapes` builds it (scale_model_per_body: every body its own length scale, s ~ U(0.85, 1.15),
    default_rng(7)): four differently proportioned humanoids + the stock one in one launch, GPU vs oracle over 10 control steps."""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import kinematics_np, quat_to_mat, scale_model_per_body
    rng = np.random.default_rng(7)
    models = [model] + [scale_model_per_body(model, np.r_[1.0, rng.uniform(0.85, 1.15, size=model.nbody - 1)]) for _ in range(4)]

    def lowest(m, q):
        xp, xq, _, _ = kinematics_np(m, q)
        return min((m.mesh_vert[m.geom_vertadr[g]:m.geom_vertadr[g] + m.geom_vertnum[g]] @ quat_to_mat(xq[m.geom_bodyid[g]]).T + xp[m.geom_bodyid[g]])[:, 2].min()
                   for g in range(m.ngeom) if m.geom_type[g] == 7)

    env_model = [0, 1, 2, 3, 4, 2, 4, 1]
    n = len(env_model)
    qpos, qvel = _states(standing, model, n, 11, noise=0.03, vel=0.1)
    for e in range(n):  # a longer-legged body carries its root higher: keep the feet where the stock model has them
        qpos[e, 2] += lowest(model, qpos[e]) - lowest(models[env_model[e]], qpos[e])
    act = np.random.default_rng(12).normal(scale=0.1, size=(n, ctrl.action_dim))
    b = S.SimBatch(models, ctrl, n, env_model=env_model)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
    os_ = [OracleSim(models[env_model[e]], ctrl) for e in range(n)]
    for e in range(n):
        os_[e].set_state(qpos[e], qvel[e])
    ncon = 0
    for t in range(10):
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        redo = b.field(S.F_REDO).cpu().numpy()
        for e in range(n):
            os_[e].do_simulation(act[e], qpos[e, 7:], redo=redo[e])
            ncon += os_[e].geti("ncon")
    gq, gv, gm = b.field(S.F_QPOS).cpu().numpy(), b.field(S.F_QVEL).cpu().numpy(), b.field(S.F_QM).cpu().numpy()
    assert ncon > 0  # the feet are on the ground: the per-model hull vertices and invweight0 constants are exercised, not only the tree
    for e in range(n):
        np.testing.assert_allclose(gq[e], os_[e].get("qpos"), atol=1e-9)
        np.testing.assert_allclose(gv[e], os_[e].get("qvel"), atol=1e-7)
        np.testing.assert_allclose(gm[e], os_[e].get("qM"), atol=1e-9)
    assert np.abs(gm[1] - gm[0]).max() > 1e-3  # different bodies indeed


def test_generated_shape_models_share_one_batch(model, ctrl, standing, kernel_path):
    """(f)-1 end to end: beta -> vertices (synthetic body provider: the SMPL files are licensed) -> per-joint hulls -> MJCF -> compiled
    model, for three betas whose hulls differ in size (`common_mesh_layout` pads them to one vertex layout; the hull graphs travel in the
    model blobs).  The generated hinge models self-collide and carry the rel_joint_lm ranges.  GPU vs oracle over 20 control steps, every
    env on its own model, as the reference rebuilds its model at every load_expert (uhc/envs/humanoid_im.py:154-190)."""
    import dataclasses
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.data_loaders.synthetic import make_synthetic_body_provider
    from uhc_amd.model.mjcf import kinematics_np, quat_to_mat
    from uhc_amd.smpllib.smpl_robot import generate_shape_models
    clips = {f"c{i}": dict(beta=np.array([0.4 * i, -0.3 * i, 0.3 * i] + [0.0] * 7), gender=i % 3) for i in range(3)}
    models, cm = generate_shape_models({"mesh": True, "model": "smpl"}, clips, make_synthetic_body_provider())
    models = [dataclasses.replace(m, solver=1) for m in models]
    assert len(models) == 3 and all((m.nq, m.nv, m.nu) == (76, 75, 69) for m in models)
    assert all((np.asarray(m.geom_contype)[1:] == 1).all() and m.nexclude == 2 for m in models)
    assert any(np.diff(m.mesh_adjadr).min() == 0 for m in models)  # some hull was padded: the layouts really differed
    assert not np.array_equal(models[0].mesh_adj, models[2].mesh_adj)

    def lowest(m, q):
        xp, xq, _, _ = kinematics_np(m, q)
        return min((m.mesh_vert[m.geom_vertadr[g]:m.geom_vertadr[g] + m.geom_vertnum[g]] @ quat_to_mat(xq[m.geom_bodyid[g]]).T + xp[m.geom_bodyid[g]])[:, 2].min()
                   for g in range(m.ngeom) if m.geom_type[g] == 7)

    env_model = [0, 1, 2, 2, 1]
    n = len(env_model)
    qpos, qvel = _states(standing, model, n, 21, noise=0.03, vel=0.1)
    for e in range(n):
        qpos[e, 2] += 0.002 - lowest(models[env_model[e]], qpos[e])  # feet 2 mm above the floor whatever the stature
    act = np.random.default_rng(22).normal(scale=0.1, size=(n, ctrl.action_dim))
    b = S.SimBatch(models, ctrl, n, env_model=env_model)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
    os_ = [OracleSim(models[env_model[e]], ctrl) for e in range(n)]
    for e in range(n):
        os_[e].set_state(qpos[e], qvel[e])
    worst, ncon = 0.0, 0
    for t in range(20):
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        redo = b.field(S.F_REDO).cpu().numpy()
        gq = b.field(S.F_QPOS).cpu().numpy()
        for e in range(n):
            os_[e].do_simulation(act[e], qpos[e, 7:], redo=redo[e])
            ncon += os_[e].geti("ncon")
            worst = max(worst, np.abs(gq[e] - os_[e].get("qpos")).max())
    assert ncon > 0 and worst < 1e-6, (ncon, worst)
    assert int(b.field(S.F_FAIL).sum().item()) == 0 and int(b.field(S.F_EFC_OVERFLOW).sum().item()) == 0


def test_oracle_decides_its_own_solver(model, ctrl, standing, kernel_path):
    """The other trajectory tests hand the device's UHC_F_REDO word to the oracle, which then takes the sweeps in exactly the substeps the
    device swept -- a wrong fallback decision on the device would be invisible there.  Here nothing is handed over: the oracle runs its own
    exact solve in every substep, free-running for 60 control steps, on a scene that stays within every tier's capacity (a few floor
    contacts); the device must not have swept anywhere (bit 1 clear), whichever tier computed the step, and the two must agree."""
    import dataclasses
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    model = dataclasses.replace(model, solver=1)
    n = 4
    qpos, qvel = _states(standing, model, n, 14, noise=0.02, vel=0.05)
    rng = np.random.default_rng(15)
    b = _sim(model, ctrl, n)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
    os_ = [OracleSim(model, ctrl) for _ in range(n)]
    for e in range(n):
        os_[e].set_state(qpos[e], qvel[e])
    worst, swept = 0.0, 0
    for t in range(60):
        act = rng.normal(scale=0.05, size=(n, ctrl.action_dim))
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        gq, gv = b.field(S.F_QPOS).cpu().numpy(), b.field(S.F_QVEL).cpu().numpy()
        redo = b.field(S.F_REDO).cpu().numpy()
        swept += int(((redo & 2) != 0).sum())
        assert ((redo & 1) != 0).all() == (kernel_path == "general") or kernel_path == "fast"
        for e in range(n):
            os_[e].do_simulation(act[e], qpos[e, 7:])  # no redo= : the checker's own decision
            worst = max(worst, np.abs(gq[e] - os_[e].get("qpos")).max(), np.abs(gv[e] - os_[e].get("qvel")).max())
    assert swept == 0, swept
    assert worst < 1e-8, worst


def test_step_inside_a_hip_graph_replays_the_eager_step(model, ctrl, standing, kernel_path):
    """`uhc_batch_simulate` captured into a HIP graph (torch.cuda.graph) with sticky tiers selected: the sticky launch reads queue
    lengths on the host between steps, which a capture cannot do, so a captured step takes the plain tier chain -- and a replay computes
    what the eager step computes, from whatever state the batch is in when it is replayed."""
    import torch
    from uhc_amd import sim as S
    n = 32
    qpos, qvel = _states(standing, model, n, 23)
    a = torch.from_numpy(np.random.default_rng(5).normal(scale=0.2, size=(n, ctrl.action_dim))).cuda()
    tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
    eager, graphed = _sim(model, ctrl, n), _sim(model, ctrl, n)
    for bb in (eager, graphed):
        bb.set_kernel_path(2)
        bb.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
        bb.sync()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        graphed.use_current_stream()
        graphed.simulate(a, tb)  # (warm-up on the capture stream, as torch asks for)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            graphed.use_current_stream()
            graphed.simulate(a, tb)
        g.replay()
        g.replay()
        side.synchronize()
    for _ in range(3):
        eager.simulate(a, tb)
    eager.sync()
    np.testing.assert_allclose(graphed.field(S.F_QPOS).cpu().numpy(), eager.field(S.F_QPOS).cpu().numpy(), atol=1e-9)
    assert int(graphed.field(S.F_FAIL).sum().item()) == 0
