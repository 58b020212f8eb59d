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


# ---------------------------------------------------------------------------------------------------------------
# Behavioural known-answer tests of the contact model and the integrator.  MuJoCo itself is absent, so these do
# not pin the restatement to MuJoCo's numbers; they make it falsifiable: a mis-stated friction cone, reference acceleration,
# regulariser or contact placement fails them.  tests/test_gpu_physics.py runs the same scenarios on the HIP path.
# ---------------------------------------------------------------------------------------------------------------
def scenario_tilted_gravity_box(tan_theta, hx=0.1, hz=0.02):
    """Flat box (tumbles only beyond tan(theta) = hx / hz = 5) resting on the plane, gravity tilted by theta about y -- the same as
    tilting the plane: returns (model, qpos0)."""
    from uhc_amd.model.mjcf import compile_mjcf
    from tests.helpers import BOX_ON_PLANE_XML, box_triangles
    m = compile_mjcf(BOX_ON_PLANE_XML, meshes={"box": box_triangles(hx, hx, hz)})
    th = np.arctan(tan_theta)
    m.gravity = 9.81 * np.array([np.sin(th), 0.0, -np.cos(th)])
    m.solver = 1
    return m, np.array([0, 0, hz - 0.0003, 1, 0, 0, 0.0])


@pytest.mark.parametrize("tan_theta,slides", [(0.8, False), (0.95, False), (1.1, True), (1.5, True)])
def test_friction_cone_slip_threshold(tan_theta, slides):
    """Coulomb threshold of the pyramidal cone along a pyramid axis: with mu = 1 (MuJoCo's default sliding friction) a box holds on
    a slope of tan(theta) < mu and slides with a = g (sin(theta) - mu cos(theta)) beyond it.  The soft constraint lets a held box
    creep (regularised friction), which bounds the 'holds' side from below."""
    from oracle.physics import OracleSim
    m, q0 = scenario_tilted_gravity_box(tan_theta)
    s = OracleSim(m)
    s.set_state(q0, np.zeros(6))
    n = 500
    for _ in range(n):
        s.step()
    t = n * m.timestep
    x = s.get("qpos")[0]
    th = np.arctan(tan_theta)
    assert s.geti("fail") == 0 and abs(s.get("qpos")[2] - 0.02) < 1e-2  # still on the plane (a sliding box chatters on its corners)
    if slides:
        a = 9.81 * (np.sin(th) - 1.0 * np.cos(th))
        assert x == pytest.approx(0.5 * a * t * t, rel=0.12)
    else:
        assert abs(x) < 0.02 * 0.5 * 9.81 * np.sin(th) * t * t  # < 2 % of the frictionless slide


def scenario_pushed_box(force_frac, hx=0.05, hz=0.2):
    """Tall box (half extents hx, hx, hz) on the plane, pushed horizontally at its centre: tips iff F hz > m g hx."""
    from uhc_amd.model.mjcf import compile_mjcf
    from tests.helpers import BOX_ON_PLANE_XML, box_triangles
    m = compile_mjcf(BOX_ON_PLANE_XML, meshes={"box": box_triangles(hx, hx, hz)})
    m.solver = 1
    f = force_frac * m.body_mass[1] * 9.81 * hx / hz
    return m, np.array([0, 0, hz - 0.0003, 1, 0, 0, 0.0]), np.array([f, 0, 0, 0, 0, 0.0])


@pytest.mark.parametrize("force_frac,tips", [(0.8, False), (1.2, True)])
def test_tipping_threshold_of_a_pushed_box(force_frac, tips):
    """Torque balance of a multi-point support: the normal forces can shift to the leading edge and no further, so a horizontal push
    at the centre of mass tips the box iff F h > m g a (friction mu = 1 > a / h keeps it from sliding first)."""
    from oracle.physics import OracleSim
    m, q0, push = scenario_pushed_box(force_frac)
    s = OracleSim(m)
    s.set_state(q0, np.zeros(6))
    for _ in range(100):  # settle
        s.step()
    s.set("qfrc_applied", push)
    for _ in range(400):
        s.step()
    w = s.get("qpos")[3]
    tilt = 2 * np.arccos(min(1.0, abs(w)))
    assert (tilt > 0.5) if tips else (tilt < 0.02)


def test_resting_penetration_matches_the_solref_solimp_equilibrium():
    """At rest (v = 0, J qacc = 0) every active pyramid edge carries f = (1 / R_py) k imp(r) (margin - r) with R_py = 2 mu^2 R_0,
    R_0 = (1 - imp) / imp (1 + mu^2) (invweight_body + invweight_world) [MJ-ext mj_makeImpedance], k = 1 / (dmax^2 tc^2 dr^2); the edge forces add up to
    the weight.  Solve that scalar equation for the penetration independently and compare with where the box actually rests."""
    from oracle.physics import OracleSim
    from uhc_amd.model.mjcf import compile_mjcf
    from tests.helpers import BOX_ON_PLANE_XML, prism_triangles
    m = compile_mjcf(BOX_ON_PLANE_XML, meshes={"box": prism_triangles(0.1, 0.05)})  # rests level on its three bottom corners
    m.solver = 1
    s = OracleSim(m)
    s.set_state(np.array([0, 0, 0.0499, 1, 0, 0, 0.0]), np.zeros(6))
    for _ in range(1500):
        s.step()
    s.forward()
    assert abs(s.get("qvel")).max() < 1e-6 and s.geti("ncon") == 3
    f = s.get("efc_force")
    dist = s.get("con_dist")
    ncon = s.geti("ncon")
    active = f.reshape(ncon, 4).sum(1) > 0
    assert f.sum() == pytest.approx(m.body_mass[1] * 9.81, rel=1e-6)  # every edge Jacobian has a unit normal component
    mu, margin, tc, dr = 1.0, 0.001, 0.02, 1.0
    dmin, dmax, width, mid, power = 0.9, 0.95, 0.001, 0.5, 2.0
    k = 1.0 / (dmax * dmax * tc * tc * dr * dr)
    tran = m.body_invweight0[1, 0] + m.body_invweight0[0, 0]

    def imp(r):
        x = abs(r - margin) / width
        if x >= 1:
            return dmax
        y = (x / mid) ** power * mid if x <= mid else 1 - ((1 - x) / (1 - mid)) ** power * (1 - mid)
        return dmin + y * (dmax - dmin)

    def edge_force(r):
        i = imp(r)
        R0 = (1 - i) / i * (tran + mu * mu * tran)
        return k * i * (margin - r) / (2 * mu * mu * R0)

    # the box rests level on its active corners: all share one distance r; 4 edges each
    r_act = dist[active]
    assert np.ptp(r_act) < 1e-6
    n_edges = 4 * int(active.sum())
    lo, hi = -0.01, margin
    for _ in range(200):  # bisection on the monotone force law
        mid_r = 0.5 * (lo + hi)
        if n_edges * edge_force(mid_r) > m.body_mass[1] * 9.81:
            lo = mid_r
        else:
            hi = mid_r
    assert r_act.mean() == pytest.approx(0.5 * (lo + hi), abs=1e-7)


def test_flat_box_contacts_are_its_bottom_hull_vertices():
    """Plane vs convex mesh: the contacts of a box lying flat are bottom corners of the hull (support vertex + hull-graph neighbours
    inside the margin), placed half-way between vertex and plane, with the plane's normal."""
    from oracle.physics import OracleSim
    m = box_model(0.1)
    s = OracleSim(m)
    s.set_state(np.array([0.3, -0.2, 0.1004, 1, 0, 0, 0.0]), np.zeros(6))
    ncon = s.geti("ncon")
    assert ncon in (3, 4)  # 4 when the triangulated bottom face's diagonal starts at the support corner
    pos = s.get("con_pos").reshape(ncon, 3)
    frame = s.get("con_frame").reshape(ncon, 9)
    corners = {(round(0.3 + sx * 0.1, 6), round(-0.2 + sy * 0.1, 6)) for sx in (-1, 1) for sy in (-1, 1)}
    assert {(round(p[0], 6), round(p[1], 6)) for p in pos} <= corners and len({tuple(p.round(6)) for p in pos}) == ncon
    np.testing.assert_allclose(pos[:, 2], 0.5 * 0.0004, atol=1e-12)
    np.testing.assert_allclose(s.get("con_dist"), 0.0004, atol=1e-12)
    np.testing.assert_allclose(frame[:, :3], np.tile([0, 0, 1.0], (ncon, 1)), atol=1e-15)


def test_implicit_joint_damping_is_backward_euler_in_the_velocity():
    """[MJ-ext] mj_Euler with dof_damping > 0 solves (M + h B) a = f: a spinning damped hinge without gravity then decays by exactly
    I / (I + h b) per step (backward Euler), while the reported qacc stays the explicit -b w / I."""
    from oracle.physics import OracleSim
    m = pendulum_model(length=0.5, half=0.05)
    m.gravity = np.zeros(3)
    b = 0.7
    m.dof_damping = np.array([b])
    s = OracleSim(m)
    I = 1.0 / m.dof_invweight0[0]
    w0, h, n = 3.0, m.timestep, 400
    s.set_state(np.array([0.1]), np.array([w0]))
    assert s.get("qacc")[0] == pytest.approx(-b * w0 / I, rel=1e-12)
    for _ in range(n):
        s.step()
    assert s.get("qvel")[0] == pytest.approx(w0 * (I / (I + h * b)) ** n, rel=1e-11)
    assert s.get("qvel")[0] != pytest.approx(w0 * (1 - h * b / I) ** n, rel=1e-6)  # not the explicit update


# ---------------------------------------------------------------------------------------------------------------
# Convex-convex narrow phase (MPR) and contacts between two moving bodies
# ---------------------------------------------------------------------------------------------------------------
def test_mpr_depth_normal_and_position_of_overlapping_boxes():
    """Two axis-aligned cubes overlapping by d along x: penetration depth d (+ margin through the inflated supports), normal +x from
    geom 1 to geom 2, contact point in the middle of the overlap slab."""
    from oracle.physics import OracleSim
    from tests.helpers import two_box_model
    m = two_box_model(0.1, 0.06)
    s = OracleSim(m)
    d = 0.004
    # A at the origin height 1 (far from the floor), B shifted along +x so that the faces overlap by d, offset in y/z inside A's face
    q = np.array([0, 0, 1.0, 1, 0, 0, 0, 0.1 + 0.06 - d, 0.013, 1.0 + 0.021, 1, 0, 0, 0.0])
    s.set_state(q, np.zeros(12))
    assert s.geti("ncon") == 1
    np.testing.assert_allclose(s.get("con_dist"), [-d], atol=2e-6)       # dist = margin - depth(inflated) = -d, to the MPR tolerance
    np.testing.assert_allclose(s.get("con_frame")[:3], [1, 0, 0], atol=1e-6)
    pos = s.get("con_pos")
    # libccd places the point by the barycentric coordinates of the origin in the final portal TETRAHEDRON, whose apex is the pair of
    # hull centres: the point sits a fraction depth / |centre distance| behind the overlap slab's middle (0.098), towards the centres
    assert 0.1 - d / 2 - 0.001 < pos[0] <= 0.1 - d / 2 + 1e-6 and abs(pos[1] - 0.013) < 0.06 and abs(pos[2] - 1.021) < 0.06
    # separated by more than the margin: no contact; inside the margin: a contact with positive distance
    q[7] = 0.16 + 0.002
    s.set_state(q, np.zeros(12))
    assert s.geti("ncon") == 0
    q[7] = 0.16 + 0.0004
    s.set_state(q, np.zeros(12))
    assert s.geti("ncon") == 1 and s.get("con_dist")[0] == pytest.approx(0.0004, abs=2e-6)


def test_mpr_is_rotation_and_translation_covariant():
    from oracle.physics import OracleSim
    from tests.helpers import two_box_model
    from scipy.spatial.transform import Rotation as sR
    m = two_box_model(0.1, 0.06)
    s = OracleSim(m)
    rng = np.random.default_rng(5)
    qa = sR.from_rotvec(rng.normal(size=3) * 0.4)
    qb = sR.from_rotvec(rng.normal(size=3) * 0.7)
    pa, pb = np.array([0.0, 0, 2.0]), np.array([0.12, 0.05, 2.03])

    def run(Rw, tw):
        A, B = Rw * qa, Rw * qb
        xa, xb = Rw.apply(pa) + tw, Rw.apply(pb) + tw
        q = np.r_[xa, np.roll(A.as_quat(), 1), xb, np.roll(B.as_quat(), 1)]
        s.set_state(q, np.zeros(12))
        assert s.geti("ncon") == 1
        return s.get("con_dist")[0], s.get("con_frame")[:3].copy(), s.get("con_pos").copy()

    d0, n0, p0 = run(sR.identity(), np.zeros(3))
    assert d0 < 0
    Rw, tw = sR.from_rotvec([0.3, -1.1, 0.5]), np.array([0.4, -0.2, 3.0])
    d1, n1, p1 = run(Rw, tw)
    assert d1 == pytest.approx(d0, abs=1e-6)
    np.testing.assert_allclose(n1, Rw.apply(n0), atol=1e-5)
    np.testing.assert_allclose(p1, Rw.apply(p0) + tw, atol=1e-5)


def test_mpr_on_round_hulls_finds_the_overlap_along_the_centre_line():
    """Two 52-vertex polyhedral 'spheres' (Fibonacci points + the two poles of the centre line) overlapping by d: the penetration
    direction is the centre line (to the faceting, a few degrees), the depth is d (the poles are exact support points along it), the
    contact point lies on the centre line inside the lens -- for three directions in space."""
    from oracle.physics import OracleSim
    from tests.helpers import TWO_BOX_XML, hull_triangles
    from uhc_amd.model.mjcf import compile_mjcf

    def ball(r, u, n=50):
        k = np.arange(n) + 0.5
        phi, th = np.arccos(1 - 2 * k / n), np.pi * (1 + 5 ** 0.5) * k
        p = np.c_[np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)]
        p = p[np.abs(p @ u) < 0.995]  # make room for the exact poles
        return r * np.r_[p, [u], [-u]]

    ra, rb, d = 0.1, 0.07, 0.012
    for u in (np.array([1.0, 0, 0]), np.array([0.0, 0.6, 0.8]), np.array([2.0, -1.0, 2.0]) / 3.0):
        m = compile_mjcf(TWO_BOX_XML, meshes={"a": hull_triangles(ball(ra, u)), "b": hull_triangles(ball(rb, u))})
        s = OracleSim(m)
        ca = np.array([0.0, 0.0, 2.0])
        cb = ca + (ra + rb - d) * u
        s.set_state(np.r_[ca, 1, 0, 0, 0, cb, 1, 0, 0, 0], np.zeros(12))
        assert s.geti("ncon") == 1
        nrm, pos, dist = s.get("con_frame")[:3], s.get("con_pos"), s.get("con_dist")[0]
        assert nrm @ u > np.cos(np.deg2rad(8.0)), (u, nrm)            # from geom 1 to geom 2, along the centre line
        assert -d - 1e-5 <= dist <= -0.8 * d, (u, dist)               # never deeper than the true overlap; the facets may hide a little of it
        t = (pos - ca) @ u
        assert ra - d - 1e-3 <= t <= ra + 1e-3 and np.linalg.norm(pos - ca - t * u) < 0.03, (u, pos)


def test_box_stacked_on_box_rests_and_carries_its_weight():
    """Two trees, a two-body contact row: the small box rests on the big one, the floor carries both weights, the box-box contact
    carries the upper weight (Newton's third law through J_b2 - J_b1)."""
    from oracle.physics import OracleSim
    from tests.helpers import two_box_model
    m = two_box_model(0.1, 0.06)
    m.solver = 1
    s = OracleSim(m)
    s.set_state(np.array([0, 0, 0.0998, 1, 0, 0, 0, 0.01, -0.02, 0.2 + 0.0598, 1, 0, 0, 0.0]), np.zeros(12))
    fc = np.zeros(12)
    for t in range(1500):
        s.step()
        if t >= 1000:
            fc += s.get("qfrc_constraint") / 500
    # MPR yields ONE contact point per hull pair, so the upper box keeps rocking gently on it (as mesh-on-mesh does in MuJoCo 2.1)
    assert s.geti("fail") == 0 and abs(s.get("qvel")).max() < 0.5
    q = s.get("qpos")
    # (the single contact sits at a corner of the face and hops between corners: the box settles ~8 mm deep instead of < 1 mm)
    assert q[2] == pytest.approx(0.1, abs=2e-3) and 0.245 < q[9] < 0.262 and np.hypot(q[7] - 0.01, q[8] + 0.02) < 0.02
    wa, wb = m.body_mass[1] * 9.81, m.body_mass[2] * 9.81
    assert fc[8] == pytest.approx(wb, rel=3e-2)          # the upper box: held up by its weight (time average)
    assert fc[2] == pytest.approx(wa, rel=3e-2)          # the lower box: floor (wa + wb) minus the load from above (wb)


def test_self_collision_variant_generates_body_body_contacts(model, standing):
    from oracle.physics import OracleSim
    from uhc_amd.model.mjcf import self_collision_variant
    sc = self_collision_variant(model)
    assert sc.nexclude == 2 and (sc.geom_contype[1:] == 1).all()
    a, b = OracleSim(model), OracleSim(sc)
    a.set_state(standing["qpos"], np.zeros(75))
    b.set_state(standing["qpos"], np.zeros(75))
    n0, n1 = a.geti("ncon"), b.geti("ncon")
    assert n1 > n0  # the static asset's arm hulls touch the torso in the standing pose
    # every new contact separates two hulls: normal force along the normal does no work on the common ancestors' dofs
    J = b.get("efc_J").reshape(b.geti("nefc"), 75)
    nl = b.geti("nefc") - (4 * n0 + (n1 - n0))
    two = J[nl + 4 * n0:]
    assert two.shape[0] == n1 - n0 and np.abs(two[:, :6]).max() < 1e-12  # internal forces: no net wrench on the root


# ---------------------------------------------------------------------------------------------------------------
# Ball joints (robot.ball: True -- config/copycat_ball/copycat_ball_1.yml:99) and free objects
# ---------------------------------------------------------------------------------------------------------------
def test_ball_variant_equals_hinge_model_where_the_coordinates_coincide(model, standing):
    """At zero joint angles the hinge rates (z, y, x) of a body ARE its body-frame angular velocity (x, y, z reversed), so the ball
    model's mass matrix and bias forces must equal the hinge model's under that permutation -- kinematics, cdof, CRB and RNE of ball
    joints against the (pinned-by-FK, KAT-tested) hinge path.  Away from zero the poses still agree through the quaternion map."""
    from oracle.physics import OracleSim
    from uhc_amd.model.mjcf import ball_variant, hinge_to_ball_qpos
    ball = ball_variant(model)
    assert (ball.nq, ball.nv, ball.nu, ball.njnt) == (99, 75, 69, 24) and ball.dof_madr[-1] == model.dof_madr[-1]
    q = model.qpos0.copy()
    q[2] = 5.0
    q[3:7] = standing["qpos"][3:7]
    perm = np.r_[np.arange(6), [6 + 3 * b + (2 - k) for b in range(23) for k in range(3)]]  # ball dof (b, x/y/z) <- hinge dof (b, z/y/x)
    h, s = OracleSim(model), OracleSim(ball)
    h.set_state(q, np.zeros(75))
    s.set_state(hinge_to_ball_qpos(model, ball, q), np.zeros(75))
    np.testing.assert_allclose(s.full_m(), h.full_m()[np.ix_(perm, perm)], atol=1e-12)
    np.testing.assert_allclose(s.get("qfrc_bias"), h.get("qfrc_bias")[perm], atol=1e-10)
    np.testing.assert_allclose(s.get("xpos"), h.get("xpos"), atol=1e-14)
    # a bent pose: same body poses through the quaternion map, M still symmetric positive definite with the total mass on the root
    qb = standing["qpos"].copy()
    qb[2] = 5.0
    h.set_state(qb, np.zeros(75))
    s.set_state(hinge_to_ball_qpos(model, ball, qb), np.zeros(75))
    np.testing.assert_allclose(s.get("xpos"), h.get("xpos"), atol=1e-12)
    np.testing.assert_allclose(s.get("xquat"), h.get("xquat"), atol=1e-12)
    M = s.full_m()
    assert np.linalg.eigvalsh(M).min() > 0.009 and M[0, 0] == pytest.approx(model.body_mass.sum(), rel=1e-12)
    # free fall: the centre of mass accelerates with g whatever the joint type
    com_acc = (M @ s.get("qacc"))[:3] / model.body_mass.sum()
    np.testing.assert_allclose(com_acc, [0, 0, -9.81], atol=1e-9)


def test_ball_motor_gear_is_a_torque_vector_in_the_child_frame(model, standing):
    """[MJ-ext] joint transmission on a ball joint: qfrc_actuator[dofs] = gear * ctrl; with gear = the old hinge axis, driving motor k of a
    body at the zero pose produces the acceleration the hinge model gets from the same torque on hinge k."""
    from oracle.physics import OracleSim
    from uhc_amd.model.mjcf import ball_variant, hinge_to_ball_qpos
    ball = ball_variant(model)
    q = model.qpos0.copy()
    q[2] = 5.0
    h, s = OracleSim(model), OracleSim(ball)
    ctrl = np.zeros(69)
    ctrl[[4, 30, 61]] = [20.0, -15.0, 7.0]
    perm = np.r_[np.arange(6), [6 + 3 * b + (2 - k) for b in range(23) for k in range(3)]]
    for sim, qq in ((h, q), (s, hinge_to_ball_qpos(model, ball, q))):
        sim.set_state(qq, np.zeros(75))
        sim.set("ctrl", ctrl)
        sim.forward()
    np.testing.assert_allclose(s.get("qfrc_actuator"), h.get("qfrc_actuator")[perm], atol=1e-14)
    np.testing.assert_allclose(s.get("qacc"), h.get("qacc")[perm], atol=1e-8)


def test_ball_joint_free_flight_conserves_angular_momentum(model, standing):
    from oracle.physics import OracleSim
    from uhc_amd.model.mjcf import ball_variant, hinge_to_ball_qpos
    errs = []
    for div in (1, 2):
        ball = ball_variant(model)
        ball.gravity = np.zeros(3)
        ball.timestep = model.timestep / div
        s = OracleSim(ball)
        rng = np.random.default_rng(8)
        qb = standing["qpos"].copy()
        qb[2] = 5.0
        s.set_state(hinge_to_ball_qpos(model, ball_variant(model), qb), rng.normal(scale=1.0, size=75))

        def ang_mom():
            s.forward()
            # spatial momentum of the whole tree about its COM = sum_b cinert_b * cvel_b (angular part); both are expressed at the tree COM
            ci, cv = s.get("cinert").reshape(-1, 10), s.get("cvel").reshape(-1, 6)
            L = np.zeros(3)
            for b in range(1, ball.nbody):
                I = np.array([[ci[b, 0], ci[b, 3], ci[b, 4]], [ci[b, 3], ci[b, 1], ci[b, 5]], [ci[b, 4], ci[b, 5], ci[b, 2]]])
                L += I @ cv[b, :3] + np.cross(ci[b, 6:9], cv[b, 3:])
            return L

        L0 = ang_mom()
        for _ in range(40 * div):
            s.step()
        errs.append(np.abs(ang_mom() - L0).max() / np.abs(L0).max())
    assert errs[0] < 2e-2 and errs[0] / errs[1] == pytest.approx(2.0, rel=0.25)  # first-order integrator: the drift halves with the step


BALL_PENDULUM_XML = """
<mujoco>
  <compiler angle="radian" coordinate="local" inertiafromgeom="true"/>
  <option timestep="0.0005"/>
  <default><geom contype="0" conaffinity="0" margin="0.001"/></default>
  <asset><mesh name="bob" file="unused.stl"/></asset>
  <worldbody>
    <body name="arm" pos="0 0 2">
      <joint name="ball" type="ball" pos="0 0 0" limited="true" range="0 0.5"/>
      <geom type="mesh" mesh="bob"/>
    </body>
  </worldbody>
</mujoco>
"""


def test_ball_joint_limit_row_and_cone():
    """[MJ-ext] mj_instantiateLimit, ball joint: value = the rotation angle of the joint's quaternion, dist = max(range) - angle, ONE row
    with Jacobian -axis on the three dofs, active when dist < margin; the soft limit then keeps a swinging ball pendulum inside its cone."""
    from oracle.physics import OracleSim
    from tests.helpers import box_triangles
    from uhc_amd.model.mjcf import compile_mjcf
    m = compile_mjcf(BALL_PENDULUM_XML, meshes={"bob": box_triangles(0.05, 0.05, 0.05, center=(0, 0, -0.5))})
    assert (m.nq, m.nv) == (4, 3) and m.jnt_limited[0] == 1 and m.jnt_range[0] == pytest.approx([0, 0.5])
    s = OracleSim(m)
    axis = np.array([0.6, -0.48, 0.64])
    for ang, active in ((0.3, False), (0.6, True), (-0.6, True), (2 * np.pi - 0.6, True), (0.5 - 1e-4, False), (0.5 + 1e-4, True)):
        q = np.r_[np.cos(ang / 2), np.sin(ang / 2) * axis]
        s.set_state(q, np.array([0.1, 0.2, -0.3]))
        assert s.geti("nefc") == int(active), ang
        if active:
            wrapped = (ang + np.pi) % (2 * np.pi) - np.pi
            J = s.get("efc_J").reshape(-1, 3)[0]
            np.testing.assert_allclose(J, -np.sign(wrapped) * axis, atol=1e-12)
            assert s.get("efc_pos")[0] == pytest.approx(0.5 - abs(wrapped), abs=1e-12)
    # J . qvel = d(dist)/dt: integrate a tiny step and compare the change of the angle
    ang = 0.7
    q = np.r_[np.cos(ang / 2), np.sin(ang / 2) * axis]
    w = np.array([0.4, -0.1, 0.25])
    s.set_state(q, w)
    J, pos0 = s.get("efc_J").reshape(-1, 3)[0].copy(), s.get("efc_pos")[0]
    h = 1e-6
    dq = np.r_[1.0, 0.5 * h * w]  # q <- q (x) exp(h w / 2), w in the child frame
    qn = np.array([q[0] * dq[0] - q[1:] @ dq[1:], *(q[0] * dq[1:] + dq[0] * q[1:] + np.cross(q[1:], dq[1:]))])
    s.set_state(qn / np.linalg.norm(qn), w)
    assert (s.get("efc_pos")[0] - pos0) / h == pytest.approx(J @ w, abs=1e-5)
    # a pendulum thrown sideways against its 0.5 rad cone: the soft limit stops it a little beyond (impedance width), nothing blows up
    s.set_state(np.array([1.0, 0, 0, 0]), np.array([3.0, 0.75, 0.0]))
    peak = 0.0
    for _ in range(4000):
        s.step()
        qq = s.get("qpos")
        peak = max(peak, 2 * np.arctan2(np.linalg.norm(qq[1:]), abs(qq[0])))
    assert 0.5 < peak < 0.55 and s.geti("fail") == 0  # (the hinge limit above: 0.3 < peak < 0.33)


def _face_down_beside_a_raft(model, standing, K=7):
    """A humanoid laid face down into the floor beside a raft of K boxes whose corners dig into each other (tests/test_gpu_selfcollision.py):
    270-330 constraint rows in the first step, 150-250 of them carrying a force -- beyond the 256 rows of the device's first three tiers."""
    import dataclasses
    from uhc_amd.model.mjcf import add_free_bodies, quat_mul, self_collision_variant
    from uhc_amd.model.shapes import box_triangles
    from uhc_amd.sim import make_ctrl
    m = self_collision_variant(model)
    yaw = [0.06 * (-1) ** k for k in range(K)]
    poses = np.array([[1.0 + 0.305 * k, 1.0 + 0.01 * k, 0.1495, np.cos(y / 2), 0, 0, np.sin(y / 2)] for k, y in enumerate(yaw)], dtype=np.float64)
    m = add_free_bodies(m, [box_triangles(0.15, 0.15, 0.15)] * K, poses, density=5.0 / 0.027)
    m = dataclasses.replace(m, solver=1)
    ctrl = make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
    q = m.qpos0.copy()
    qh = standing["qpos"].copy()
    qh[3:7] = quat_mul(np.array([np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0]), qh[3:7])  # tipped forward: face down
    qh[0], qh[1], qh[2] = -0.8, -0.8, 0.14
    q[:76] = qh
    return m, ctrl, q, np.zeros(m.nv)


def test_primal_newton_reaches_the_dual_optimum(model, standing):
    """orc_solve_primal (Newton on MuJoCo's primal problem, the reference's default solver [MJ-ext]) against orc_solve_active_set (block
    pivoting on the dual) on identical forward passes along roll-outs of the generated model class and of the ball-joint humanoid among
    boxes: the same acceleration to 1e-10 (relative), a handful of Newton iterations from the warm start, never the iteration cap."""
    import dataclasses
    from oracle.physics import OracleSim
    from uhc_amd.model.mjcf import add_free_bodies, hinge_to_ball_qpos, ball_variant
    from uhc_amd.model.shapes import box_triangles
    from uhc_amd.sim import make_ctrl
    from uhc_amd.smpllib.smpl_robot import robot_variant
    rng = np.random.default_rng(21)
    gen = dataclasses.replace(robot_variant(model, {"mesh": True, "model": "smpl"}), solver=1)
    ball = robot_variant(model, {"mesh": True, "model": "smpl", "ball": True})
    poses = np.stack([np.r_[-0.15 + 0.75 * np.cos(a), -0.05 + 0.75 * np.sin(a), 0.16 + 0.35 * k, 1, 0, 0, 0] for k, a in enumerate(rng.uniform(0, 2 * np.pi, size=4))])
    ball = dataclasses.replace(add_free_bodies(ball, [box_triangles(0.15, 0.15, 0.15)] * 4, poses, density=5.0 / 0.027, friction=1.0, condim=1), solver=1)
    qb = ball.qpos0.copy()
    qb[:99] = hinge_to_ball_qpos(model, ball_variant(model), standing["qpos"])
    cases = [(gen, make_ctrl(gen), standing["qpos"].copy(), np.zeros(75), 0.05, standing["qpos"][7:].copy()),
             (ball, make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4), qb, np.zeros(ball.nv), 0.003, np.zeros(69))]
    for m, ctrl, q0, v0, a_sc, tb in cases:
        o, p = OracleSim(m, ctrl), OracleSim(dataclasses.replace(m, solver=2), ctrl)
        o.set_state(q0, v0)
        worst, iters, rows = 0.0, [], 0
        for t in range(90):
            for k in ("qacc_warmstart", "ctrl", "qfrc_applied"):
                p.set(k, o.get(k)) if t else None
            p.set_state(o.get("qpos"), o.get("qvel"))
            for k in ("qacc_warmstart", "ctrl", "qfrc_applied"):
                p.set(k, o.get(k))
            o.forward(); p.forward()
            qa, qp = o.get("qacc"), p.get("qacc")
            worst = max(worst, np.abs(qa - qp).max() / (1.0 + np.abs(qa).max()))
            if p.geti("nefc"):
                iters.append(p.geti("solver_iter"))
            rows = max(rows, p.geti("nefc"))
            np.testing.assert_allclose(p.get("efc_force"), o.get("efc_force"), atol=1e-8 * (1 + np.abs(o.get("efc_force")).max()))
            o.do_simulation(rng.normal(scale=a_sc, size=ctrl.action_dim), tb)
        assert rows > 100 and worst < 1e-10, (rows, worst)
        assert p.geti("primal_unconverged") == 0 and np.mean(iters) < 6 and max(iters) <= 12, (np.mean(iters), max(iters))


def test_primal_newton_on_more_rows_than_the_dual_tiers_hold(model, standing):
    """The face-down humanoid beside the seven-box raft: 270-330 rows.  solver 1 takes the primal path there by itself (more than 256 rows);
    its forces satisfy the KKT conditions of the dual QP, with the Delassus matrix formed independently in numpy from J, M and R:
    f >= 0, y = A f + b >= -tol, f . y = 0 -- and the sweeps run far beyond their tolerance approach the same forces."""
    import dataclasses
    from oracle.physics import OracleSim
    m, ctrl, q, v = _face_down_beside_a_raft(model, standing)
    o = OracleSim(m, ctrl)
    o.set_state(q, v)
    seen = 0
    for t in range(6):  # the first substeps of the fall: 330 ... 300 rows (then the pile settles below 256)
        n = o.geti("nefc")
        assert n > 256 and o.geti("primal_solves") == t + 1 and o.geti("primal_unconverged") == 0 and o.geti("efc_overflow") == 0, (t, n, o.geti("primal_solves"))
        J, R, b, f = o.get("efc_J").reshape(n, m.nv), o.get("efc_R"), o.get("efc_b"), o.get("efc_force")
        A = J @ np.linalg.solve(o.full_m(), J.T) + np.diag(R)
        y = A @ f + b
        scale = np.abs(b).max()
        assert (f >= 0).all() and y.min() > -1e-9 * scale and np.abs(f * y).max() < 1e-9 * scale * max(f.max(), 1.0), (y.min(), np.abs(f * y).max(), scale)
        assert (f > 0).sum() > 64  # more force-carrying rows than one register-resident working set holds
        np.testing.assert_allclose(o.get("qacc"), o.get("qacc_smooth") + np.linalg.solve(o.full_m(), J.T @ f), atol=1e-8 * (1 + np.abs(o.get("qacc")).max()))
        seen = max(seen, n)
        its = o.geti("solver_iter")
        o.step()
    sw = OracleSim(dataclasses.replace(m, solver=0, iterations=20000, tolerance=1e-20), ctrl)
    ex = OracleSim(m, ctrl)
    sw.set_state(q, v); ex.set_state(q, v)
    assert np.abs(sw.get("efc_force") - ex.get("efc_force")).max() < 1e-5 * (1 + np.abs(ex.get("efc_force")).max())
    print(f"face-down humanoid + raft: up to {seen} rows, Newton iterations of the last pass checked {its}")
