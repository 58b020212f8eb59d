"""Comparison with a REAL MuJoCo, wherever one is importable (`pip install mujoco` on a developer machine; it is absent from the
build image and from the GPU box, so these tests skip there -- the MuJoCo stages of the oracle stay "parity unpinned").

The reference's asset is written with `coordinate="global"`, which current MuJoCo no longer reads; the model is therefore compiled
by this build (uhc_amd/model/mjcf.py), written back in local coordinates (`export_mjcf`) and handed to MuJoCo.  Two exports are
compared: `density=1000` leaves mass / inertia / hull to MuJoCo's own compiler (checks polyhedron_mass_properties and set_const),
the plain one pins the inertial numbers so that the dynamics stages can be compared at rounding level.

[MJ-ext] every number MuJoCo produces here is the thing the oracle restates: qM, qfrc_bias, contacts, efc_aref / efc_R / efc_D and
one mj_step with the Newton solver (the reference's setting) against the oracle's exact solve of the same QP."""
import numpy as np
import pytest

mujoco = pytest.importorskip("mujoco")


def _mj(model, density=None):
    from uhc_amd.model.mjcf import export_mjcf
    mm = mujoco.MjModel.from_xml_string(export_mjcf(model, density=density))
    mm.opt.cone = mujoco.mjtCone.mjCONE_PYRAMIDAL       # MuJoCo 2.1 default; 3.x kept it, set anyway
    mm.opt.jacobian = mujoco.mjtJacobian.mjJAC_DENSE
    mm.opt.solver = mujoco.mjtSolver.mjSOL_NEWTON
    mm.opt.integrator = mujoco.mjtIntegrator.mjINT_EULER
    # implicit joint damping inside Euler stays on (mjDSBL_EULERDAMP clear): 2.1 does it and orc_euler restates it
    return mm, mujoco.MjData(mm)


def _state(standing, seed, lift=0.0, noise=0.2, vel=0.5):
    rng = np.random.default_rng(seed)
    qpos = standing["qpos"].copy()
    qpos[2] += lift
    qpos[7:] += rng.normal(scale=noise, size=69)
    return qpos, rng.normal(scale=vel, size=75)


def test_compiler_constants_equal_mujocos(model):
    mm, _ = _mj(model, density=1000.0)
    assert (mm.nq, mm.nv, mm.nu, mm.nbody) == (model.nq, model.nv, model.nu, model.nbody)
    np.testing.assert_array_equal(mm.body_parentid, model.body_parentid)
    np.testing.assert_array_equal(mm.dof_parentid, model.dof_parentid)
    np.testing.assert_array_equal(mm.dof_Madr, model.dof_madr[:-1])
    np.testing.assert_allclose(mm.body_pos, model.body_pos, atol=1e-12)
    # MuJoCo integrates the mesh as given (= its hull here, the vertices are a hull's); this build integrates the STL's own triangles
    np.testing.assert_allclose(mm.body_mass, model.body_mass, rtol=2e-3)
    np.testing.assert_allclose(mm.body_ipos, model.body_ipos, atol=2e-4)
    np.testing.assert_allclose(np.sort(mm.body_inertia, axis=1), np.sort(model.body_inertia, axis=1), rtol=5e-3, atol=1e-7)
    mm2, _ = _mj(model)  # inertial numbers pinned: the qpos0 constants must agree to rounding
    np.testing.assert_allclose(mm2.body_mass, model.body_mass, rtol=1e-14)
    np.testing.assert_allclose(mm2.dof_invweight0, model.dof_invweight0, rtol=1e-9)
    np.testing.assert_allclose(mm2.body_invweight0, model.body_invweight0, rtol=1e-9)
    assert mm2.stat.meaninertia == pytest.approx(model.meaninertia, rel=1e-10)


def test_smooth_dynamics_equal_mujocos(model, standing):
    from oracle.physics import OracleSim
    mm, md = _mj(model)
    for seed in range(3):
        qpos, qvel = _state(standing, seed, lift=5.0, noise=0.4, vel=2.0)
        md.qpos[:], md.qvel[:] = qpos, qvel
        mujoco.mj_forward(mm, md)
        o = OracleSim(model)
        o.set_state(qpos, qvel)
        M = np.zeros((mm.nv, mm.nv))
        mujoco.mj_fullM(mm, M, md.qM)
        np.testing.assert_allclose(o.get("xpos").reshape(-1, 3), md.xpos, atol=1e-12)
        np.testing.assert_allclose(o.get("xipos").reshape(-1, 3), md.xipos, atol=1e-12)
        np.testing.assert_allclose(o.get("qM"), md.qM, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(o.full_m(), M, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(o.get("qfrc_bias"), md.qfrc_bias, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(o.get("qacc"), md.qacc, rtol=1e-8, atol=1e-8)  # airborne: no constraints


@pytest.mark.parametrize("variant", ["floor", "self_collision"])
def test_contacts_and_constraint_rows_equal_mujocos(model, standing, variant):
    """The soft spots named in DESIGN.md 2: how many plane-mesh contacts per hull (cap 4 here), where they are placed, the hull graph,
    MPR's support tie-breaking, diagApprox / R / aref.  Contacts are matched by (geom pair, nearest position)."""
    from oracle.physics import OracleSim
    from uhc_amd.model.mjcf import self_collision_variant
    m = model if variant == "floor" else self_collision_variant(model)
    mm, md = _mj(m)
    for seed in range(3):
        qpos, qvel = _state(standing, 10 + seed, noise=0.05 if variant == "floor" else 0.3, vel=0.3)
        md.qpos[:], md.qvel[:] = qpos, qvel
        mujoco.mj_forward(mm, md)
        o = OracleSim(m)
        o.set_state(qpos, qvel)
        assert o.geti("ncon") == md.ncon, (o.geti("ncon"), md.ncon)
        mine = o.get("con_pos").reshape(-1, 3)
        theirs = np.array([md.contact[i].pos for i in range(md.ncon)]).reshape(-1, 3)
        for p in theirs:
            assert np.abs(mine - p).sum(axis=1).min() < 1e-7, p
        np.testing.assert_allclose(np.sort(o.get("con_dist")), np.sort([md.contact[i].dist for i in range(md.ncon)]), atol=1e-8)
        assert o.geti("nefc") == md.nefc
        np.testing.assert_allclose(np.sort(o.get("efc_R")), np.sort(md.efc_R[:md.nefc]), rtol=1e-7)
        np.testing.assert_allclose(np.sort(o.get("efc_aref")), np.sort(md.efc_aref[:md.nefc]), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(o.get("qacc"), md.qacc, rtol=1e-5, atol=1e-5)  # Newton (tol 1e-8) vs the exact optimum


def test_two_hundred_control_steps_track_mujoco(model, standing):
    """north_star: per-step qpos / qvel within 1e-4 of the MuJoCo-CPU path over 200 steps on the same seed -- the reference loop
    (compute_torque from the previous forward pass, rfc_implicit, mj_step) driven by the oracle's controller on both sides."""
    from oracle.physics import OracleSim
    from uhc_amd.sim import make_ctrl
    ctrl = make_ctrl(model)
    mm, md = _mj(model)
    rng = np.random.default_rng(5)
    qpos, qvel = standing["qpos"].copy(), np.zeros(75)
    o = OracleSim(model, ctrl)    # controller only: fed MuJoCo's own state, qM and qfrc_bias each substep
    o.set_state(qpos, qvel)
    o2 = OracleSim(model, ctrl)   # the oracle's whole loop, free-running
    o2.set_state(qpos, qvel)
    md.qpos[:], md.qvel[:] = qpos, qvel
    mujoco.mj_forward(mm, md)
    worst = 0.0
    for step in range(200):
        act = rng.normal(scale=0.05, size=ctrl.action_dim)
        for it in range(ctrl.n_substeps):
            # the torque the reference computes from MuJoCo's qM / qfrc_bias of the previous forward pass (humanoid_im.py:1014-1076)
            o.set("qpos", md.qpos.copy()); o.set("qvel", md.qvel.copy()); o.set("qM", md.qM.copy()); o.set("qfrc_bias", md.qfrc_bias.copy())
            md.ctrl[:] = o.pd_torque(act, qpos[7:], it)
            md.qfrc_applied[:] = o.rfc_implicit(act)
            mujoco.mj_step(mm, md)
        o2.do_simulation(act, qpos[7:])
        worst = max(worst, np.abs(o2.get("qpos") - md.qpos).max(), np.abs(o2.get("qvel") - md.qvel).max())
        if o2.geti("fail"):
            break
    assert worst < 1e-4, worst


def test_ball_joint_limit_rows_equal_mujocos(model, standing):
    """Round 4: mj_instantiateLimit's ball branch (one row per limited ball joint: angle of the joint's quaternion against max(range),
    Jacobian -axis on its three dofs) -- rows, distances, reference accelerations and the resulting acceleration."""
    from oracle.physics import OracleSim
    from uhc_amd.model.mjcf import JNT_BALL, ball_variant, hinge_to_ball_qpos
    ball = ball_variant(model).copy()
    isb = np.asarray(ball.jnt_type) == JNT_BALL
    ball.jnt_limited = isb.astype(np.int32)
    ball.jnt_range = np.where(isb[:, None], np.array([0.0, 0.3]), ball.jnt_range)
    mm, md = _mj(ball)
    for seed in range(3):
        qh, qvel = _state(standing, 20 + seed, lift=3.0, noise=0.15, vel=0.5)
        qpos = hinge_to_ball_qpos(model, ball, qh)
        md.qpos[:], md.qvel[:] = qpos, qvel
        mujoco.mj_forward(mm, md)
        o = OracleSim(ball)
        o.set_state(qpos, qvel)
        assert o.geti("nefc") == md.nefc and md.nefc > 0
        np.testing.assert_allclose(np.sort(o.get("efc_pos")), np.sort(md.efc_pos[:md.nefc]), atol=1e-10)
        np.testing.assert_allclose(np.sort(o.get("efc_R")), np.sort(md.efc_R[:md.nefc]), rtol=1e-7)
        np.testing.assert_allclose(np.sort(o.get("efc_aref")), np.sort(md.efc_aref[:md.nefc]), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(o.get("qacc"), md.qacc, rtol=1e-5, atol=1e-5)


def test_humanoid_with_free_objects_tracks_mujoco(model, standing):
    """Round 4: the env layer's objects (free bodies appended behind the humanoid, uhc/smpllib/smpl_robot.py:1200-1252): contacts between
    boxes, humanoid and floor through MuJoCo's own mesh collider vs this build's plane-mesh rule and MPR, 30 control steps."""
    from oracle.physics import OracleSim
    from tests.helpers import box_triangles
    from uhc_amd.model.mjcf import add_free_bodies, self_collision_variant
    from uhc_amd.sim import make_ctrl
    poses = np.array([[0.45, 0.0, 0.151, 1, 0, 0, 0], [-0.1, 0.5, 0.6, 0.9, 0.1, 0, 0]])
    m = add_free_bodies(self_collision_variant(model), [box_triangles(0.15, 0.15, 0.15), box_triangles(0.1, 0.2, 0.08)], poses, density=400.0, friction=1.0, condim=3)
    ctrl = make_ctrl(model)
    mm, md = _mj(m)
    qpos, qvel = np.r_[standing["qpos"], (poses / np.r_[np.ones(3), np.full(4, 1.0)]).ravel()], np.zeros(m.nv)
    qpos[76 + 7 + 3:76 + 14] /= np.linalg.norm(qpos[76 + 7 + 3:76 + 14])
    o = OracleSim(m, ctrl)
    o.set_state(qpos, qvel)
    oc = OracleSim(m, ctrl)
    oc.set_state(qpos, qvel)
    md.qpos[:], md.qvel[:] = qpos, qvel
    mujoco.mj_forward(mm, md)
    assert o.geti("ncon") == md.ncon
    worst = 0.0
    for step in range(30):
        act = np.zeros(ctrl.action_dim)
        for it in range(ctrl.n_substeps):
            oc.set("qpos", md.qpos.copy()); oc.set("qvel", md.qvel.copy()); oc.set("qM", md.qM.copy()); oc.set("qfrc_bias", md.qfrc_bias.copy())
            md.ctrl[:] = oc.pd_torque(act, standing["qpos"][7:], it)
            md.qfrc_applied[:] = oc.rfc_implicit(act)
            mujoco.mj_step(mm, md)
        o.do_simulation(act, standing["qpos"][7:])
        worst = max(worst, np.abs(o.get("qpos") - md.qpos).max())
    assert worst < 1e-4, worst
