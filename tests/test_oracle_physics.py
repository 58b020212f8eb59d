"""Analytic known-answer tests that pin the CPU oracle's MuJoCo-stage restatement (MuJoCo itself is
absent): mass matrix, bias forces, LDL solve, integration order, contacts, PGS."""
import numpy as np
import pytest

from tests.helpers import box_model, pendulum_model


@pytest.fixture(scope="module")
def sim(model, standing):
    from oracle.physics import OracleSim
    s = OracleSim(model)
    return s


def _lifted(standing, rng, lift=10.0):
    qpos = standing["qpos"].copy()
    qpos[2] += lift
    qpos[7:] += rng.normal(scale=0.3, size=69)
    return qpos, rng.normal(scale=2.0, size=75)


def test_mass_matrix_spd_and_equals_independent_crba(model, standing, sim):
    from uhc_amd.model.mjcf import mass_matrix_np
    rng = np.random.default_rng(0)
    qpos, qvel = _lifted(standing, rng)
    sim.set_state(qpos, qvel)
    M = sim.full_m()
    np.testing.assert_allclose(M, M.T, atol=1e-13)
    assert np.linalg.eigvalsh(M).min() > 0.009  # armature 0.01 bounds the hinge diagonal from below
    np.testing.assert_allclose(M, mass_matrix_np(model, qpos), atol=1e-12)
    assert M[0, 0] == pytest.approx(model.body_mass.sum(), rel=1e-12)  # total mass on the root translation


def test_sparse_ldl_solve_equals_dense(model, standing, sim):
    rng = np.random.default_rng(1)
    qpos, qvel = _lifted(standing, rng)
    sim.set_state(qpos, qvel)
    # qacc_smooth = M^-1 qfrc_smooth through the tree-sparse factorisation
    np.testing.assert_allclose(sim.full_m() @ sim.get("qacc_smooth"), sim.get("qfrc_smooth"), atol=1e-9)


def test_bias_force_gravity_and_coriolis_vs_finite_differences(model, standing):
    from oracle.physics import OracleSim
    rng = np.random.default_rng(2)
    qpos, qvel = _lifted(standing, rng)
    s = OracleSim(model)
    s.set_state(qpos, np.zeros(75))
    g_bias = s.get("qfrc_bias").copy()
    assert g_bias[2] == pytest.approx(model.body_mass.sum() * 9.81, rel=1e-12)

    def pe(q):
        s.set("qpos", q)
        s.call("kinematics")
        return 9.81 * (model.body_mass * s.get("xipos").reshape(-1, 3)[:, 2]).sum()

    eps = 1e-6
    for k in rng.choice(69, size=12, replace=False):
        qp, qm = qpos.copy(), qpos.copy()
        qp[7 + k] += eps
        qm[7 + k] -= eps
        assert (pe(qp) - pe(qm)) / (2 * eps) == pytest.approx(g_bias[6 + k], abs=2e-5)
    # Coriolis/centrifugal: c = Mdot v - 1/2 d(v^T M v)/dq on the hinge coordinates, gravity off
    m0 = model.copy()
    m0.gravity = np.zeros(3)
    s0 = OracleSim(m0)
    s0.set_state(qpos, qvel)
    c = s0.get("qfrc_bias").copy()

    def integ(q, v, h):
        q = q.copy()
        q[:3] += h * v[:3]
        w = v[3:6]
        n = np.linalg.norm(w)
        qr = np.r_[np.cos(h * n / 2), w / n * np.sin(h * n / 2)]
        a = q[3:7]
        q[3:7] = [a[0] * qr[0] - a[1] * qr[1] - a[2] * qr[2] - a[3] * qr[3], a[0] * qr[1] + a[1] * qr[0] + a[2] * qr[3] - a[3] * qr[2],
                  a[0] * qr[2] - a[1] * qr[3] + a[2] * qr[0] + a[3] * qr[1], a[0] * qr[3] + a[1] * qr[2] - a[2] * qr[1] + a[3] * qr[0]]
        q[7:] += h * v[6:]
        return q

    s0.set_state(integ(qpos, qvel, eps), qvel)
    Mp = s0.full_m()
    s0.set_state(integ(qpos, qvel, -eps), qvel)
    Mm = s0.full_m()
    Md = (Mp - Mm) / (2 * eps)
    assert qvel @ c == pytest.approx(0.5 * qvel @ Md @ qvel, rel=1e-6)  # power balance (includes the free joint)
    ks = rng.choice(69, size=10, replace=False)
    for k in ks:
        qp, qm = qpos.copy(), qpos.copy()
        qp[7 + k] += eps
        qm[7 + k] -= eps
        s0.set_state(qp, qvel)
        Kp = 0.5 * qvel @ s0.full_m() @ qvel
        s0.set_state(qm, qvel)
        Km = 0.5 * qvel @ s0.full_m() @ qvel
        assert (Md @ qvel)[6 + k] - (Kp - Km) / (2 * eps) == pytest.approx(c[6 + k], abs=5e-5)


def test_free_fall_and_first_order_integrator(model, standing):
    from oracle.physics import OracleSim
    rng = np.random.default_rng(3)
    qpos, qvel = _lifted(standing, rng)
    s = OracleSim(model)
    s.set_state(qpos, np.zeros(75))
    assert s.geti("ncon") == 0
    com_acc = (s.full_m() @ s.get("qacc"))[:3] / model.body_mass.sum()
    np.testing.assert_allclose(com_acc, [0, 0, -9.81], atol=1e-9)
    # semi-implicit Euler: momentum / energy errors halve with the step (gravity off, free flight)
    errs = []
    for div in (1, 2, 4):
        m0 = model.copy()
        m0.gravity = np.zeros(3)
        m0.timestep = model.timestep / div
        t = OracleSim(m0)
        t.set_state(qpos, qvel)
        p0 = (t.full_m() @ t.get("qvel"))[:3].copy()
        for _ in range(60 * div):
            t.step()
        t.forward()
        errs.append(np.abs((t.full_m() @ t.get("qvel"))[:3] - p0).max())
    assert errs[0] / errs[1] == pytest.approx(2.0, rel=0.1) and errs[1] / errs[2] == pytest.approx(2.0, rel=0.1)


def test_pendulum_period():
    from oracle.physics import OracleSim
    m = pendulum_model(length=0.5, half=0.05)
    s = OracleSim(m)
    I = 1.0 / m.dof_invweight0[0]  # inertia about the hinge incl. armature (0 here)
    T = 2 * np.pi * np.sqrt(I / (m.body_mass[1] * 9.81 * 0.5))
    s.set_state(np.array([0.01]), np.zeros(1))
    prev, crossings, t = s.get("qpos")[0], [], 0.0
    for _ in range(int(2.2 * T / m.timestep)):
        s.step()
        t += m.timestep
        cur = s.get("qpos")[0]
        if prev > 0 >= cur:
            crossings.append(t)
        prev = cur
    assert len(crossings) >= 2
    assert crossings[1] - crossings[0] == pytest.approx(T, rel=2e-3)


def test_box_rests_on_plane_with_its_weight():
    from oracle.physics import OracleSim
    m = box_model(0.1)
    m.iterations = 2000  # let PGS reach its tolerance so the KKT conditions can be checked tightly
    s = OracleSim(m)
    s.set_state(np.array([0, 0, 0.0995, 1, 0, 0, 0.0]), np.zeros(6))
    for _ in range(600):
        s.step()
    s.forward()
    # support vertex + its hull-graph neighbours inside the margin: 3 or 4 of the bottom corners, depending on
    # which corner carries the diagonal edge of the triangulated face
    assert s.geti("ncon") in (3, 4) and s.geti("nefc") == 4 * s.geti("ncon") and s.geti("fail") == 0
    assert abs(s.get("qvel")).max() < 2e-2  # a 3-point soft support keeps rocking slightly
    weight = m.body_mass[1] * 9.81
    assert s.get("qfrc_constraint")[2] == pytest.approx(weight, rel=2e-2)
    assert s.get("qpos")[2] == pytest.approx(0.1, abs=2e-3)  # soft contact: small penetration
    # dual problem structure: A symmetric PSD, forces non-negative, KKT complementarity at the solution
    n = s.geti("nefc")
    A = s.get("efc_AR").reshape(n, n)
    np.testing.assert_allclose(A, A.T, atol=1e-10)
    assert np.linalg.eigvalsh(A).min() > 0
    f, b = s.get("efc_force"), s.get("efc_b")
    res = A @ f + b
    assert (f >= 0).all() and (res[f == 0] > -1e-6).all() and np.abs(res[f > 0]).max() < 2e-3


def test_joint_limit_constraint():
    from oracle.physics import OracleSim
    from uhc_amd.model.mjcf import compile_mjcf
    from tests.helpers import PENDULUM_XML, box_triangles
    xml = PENDULUM_XML.replace('<joint name="hinge" type="hinge" axis="0 1 0" pos="0 0 0"/>',
                               '<joint name="hinge" type="hinge" axis="0 1 0" pos="0 0 0" limited="true" range="-0.3 0.3"/>')
    m = compile_mjcf(xml, meshes={"bob": box_triangles(0.05, 0.05, 0.05, center=(0, 0, -0.5))})
    s = OracleSim(m)
    s.set_state(np.array([0.29]), np.array([2.0]))
    peak = 0.0
    for _ in range(2000):
        s.step()
        peak = max(peak, s.get("qpos")[0])
    assert 0.3 < peak < 0.33  # soft limit: overshoot bounded by the impedance width scale
    assert s.geti("fail") == 0


def test_explicit_rfc_equals_virtual_work(model, standing):
    """orc_rfc_explicit (mj_applyFT through the point Jacobian) against the principle of virtual work:
    qfrc . v = sum_b f_b . d/dt(point_b) + tau_b . omega_b, with the right side from central differences of the
    numpy kinematics (uhc_amd/model/mjcf.py)."""
    from oracle.physics import OracleSim
    from uhc_amd.model.mjcf import kinematics_np, quat_mul, quat_to_mat
    from uhc_amd.sim import make_ctrl
    ctrl = make_ctrl(model, residual_force_mode="explicit")
    rng = np.random.default_rng(21)
    qpos = standing["qpos"].copy()
    qpos[7:] += rng.normal(scale=0.2, size=model.nu)
    q = rng.normal(size=4)
    qpos[3:7] = q / np.linalg.norm(q)
    o = OracleSim(model, ctrl)
    o.set_state(qpos, np.zeros(model.nv))
    action = rng.normal(scale=0.3, size=ctrl.action_dim)
    qfrc = o.rfc_explicit(action).copy()
    assert np.abs(qfrc).max() > 1.0
    vf = action[model.nu:model.nu + 24 * 9].reshape(24, 9)
    bodies = [ctrl.vf_body[i] for i in range(24)]

    def integrate(qp, v, h):  # free joint: world-frame translation, body-frame rotation; hinges: angle
        out = qp.copy()
        out[:3] += h * v[:3]
        w = v[3:6] * h
        ang = np.linalg.norm(w)
        dq = np.r_[np.cos(ang / 2), np.sin(ang / 2) * w / ang] if ang > 0 else np.array([1.0, 0, 0, 0])
        out[3:7] = quat_mul(qp[3:7], dq)
        out[7:] += h * v[6:]
        return out

    def points_and_frames(qp):
        xpos, xquat = kinematics_np(model, qp)[:2]
        R = [quat_to_mat(xquat[b]) for b in bodies]
        return np.array([xpos[b] + R[i] @ vf[i, :3] for i, b in enumerate(bodies)]), R

    _, R0 = points_and_frames(qpos)
    f = np.array([R0[i] @ (vf[i, 3:6] * ctrl.rfc_scale) for i in range(24)])
    tq = np.array([R0[i] @ (vf[i, 6:9] * ctrl.rfc_scale) for i in range(24)])
    h = 1e-6
    for _ in range(5):
        v = rng.normal(size=model.nv)
        pp, Rp = points_and_frames(integrate(qpos, v, h))
        pm, Rm = points_and_frames(integrate(qpos, v, -h))
        power = 0.0
        for i in range(24):
            pdot = (pp[i] - pm[i]) / (2 * h)
            dR = (Rp[i] - Rm[i]) / (2 * h) @ R0[i].T  # skew(omega)
            omega = np.array([dR[2, 1], dR[0, 2], dR[1, 0]])
            power += f[i] @ pdot + tq[i] @ omega
        assert qfrc @ v == pytest.approx(power, rel=1e-6, abs=1e-6)


def test_meta_pd_gains_follow_the_residual_block(model, standing):
    """The meta-PD scales sit after the residual-force block whatever its width (humanoid_im.py:1054-1060): the same joint
    residuals and gain scales must give the same torques under the implicit (6) and the explicit (24 x 9) layouts."""
    from oracle.physics import OracleSim
    from uhc_amd.sim import make_ctrl
    ci, ce = make_ctrl(model), make_ctrl(model, residual_force_mode="explicit")
    rng = np.random.default_rng(31)
    joint, meta = rng.normal(scale=0.3, size=69), rng.normal(scale=0.5, size=30)
    qpos = standing["qpos"].copy()
    qpos[7:] += rng.normal(scale=0.1, size=69)
    tq = []
    for c, nvf in ((ci, 6), (ce, 216)):
        o = OracleSim(model, c)
        o.set_state(qpos, rng.normal(size=75) * 0)
        tq.append(o.pd_torque(np.r_[joint, rng.normal(size=nvf), meta], standing["qpos"][7:], 7).copy())
    np.testing.assert_array_equal(tq[0], tq[1])


def test_active_set_solver_satisfies_kkt_exactly_and_is_path_independent(model, standing):
    """Oracle solver 1 (block principal pivoting on the dual QP): the result satisfies the KKT conditions of
    min 1/2 f'Af + f'b, f >= 0 to rounding (f >= 0, y = Af + b >= 0, f.y = 0) -- which tolerance-terminated sweeps do not --,
    agrees with Gauss-Seidel run far beyond its tolerance, and needs only a few factorisation rounds."""
    import dataclasses
    from oracle.physics import OracleSim
    from uhc_amd.sim import make_ctrl
    rng = np.random.default_rng(12)
    exact = dataclasses.replace(model, solver=1)
    sweeps = dataclasses.replace(model, solver=0, iterations=100)
    long_sweeps = dataclasses.replace(model, solver=0, iterations=40000, tolerance=1e-18)
    ctrl = make_ctrl(model)
    checked = 0
    for trial in range(6):
        qpos = standing["qpos"].copy()
        qpos[7:] += rng.normal(scale=0.1, size=69)
        qvel = rng.normal(scale=0.5, size=75)
        se, sp, sl = OracleSim(exact, ctrl), OracleSim(sweeps, ctrl), OracleSim(long_sweeps, ctrl)
        for s in (se, sp, sl):
            s.set_state(qpos, qvel)
        n = se.geti("nefc")
        if n == 0:
            continue
        A, b = se.get("efc_AR").reshape(n, n), se.get("efc_b")
        f = se.get("efc_force")
        y = A @ f + b
        scale = np.abs(b).max()
        assert (f >= 0).all() and y.min() > -1e-10 * scale and np.abs(f * y).max() < 1e-10 * scale * max(f.max(), 1.0)
        assert 1 <= se.geti("solver_iter") <= 10
        # sweeps at MuJoCo's defaults stop short of the optimum; sweeps run to exhaustion reach it
        fp, fl = sp.get("efc_force"), sl.get("efc_force")
        assert np.abs(fl - f).max() < 1e-7 * (1 + np.abs(f).max())
        yp = A @ fp + b
        assert np.abs(fp * yp).max() > 1e-9 * scale or np.abs(fp - f).max() > 1e-7
        np.testing.assert_allclose(sl.get("qacc"), se.get("qacc"), atol=1e-6)
        checked += 1
    assert checked >= 4
