"""GPU twins of the behavioural known-answer tests (tests/test_oracle_physics.py): the same scenarios -- friction-cone slip
threshold, tipping threshold, resting-penetration equilibrium, flat-box contacts, implicit joint damping -- on the HIP path through
the C-ABI, each also compared with the oracle's trajectory.  Small generic models (one free body / one hinge, no actuators), which
the humanoid parity tests never exercise."""
import os

import numpy as np
import pytest

from tests.helpers import passive_ctrl
from tests.test_oracle_physics import scenario_pushed_box, scenario_tilted_gravity_box

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["fast", "general"], autouse=True)
def kernel_path(request):
    old = os.environ.get("UHC_FORCE_GENERAL")
    os.environ["UHC_FORCE_GENERAL"] = "1" if request.param == "general" else "0"
    yield request.param
    if old is None:
        os.environ.pop("UHC_FORCE_GENERAL", None)
    else:
        os.environ["UHC_FORCE_GENERAL"] = old


def _run(models, q0s, v0s, n_steps, applied=None, settle=0):
    """Step a batch (one env per (model, state)) n_steps single physics steps; returns the SimBatch."""
    import torch
    from uhc_amd import sim as S
    n = len(q0s)
    b = S.SimBatch(models, passive_ctrl(models[0]), n, env_model=list(range(len(models))) if len(models) > 1 else None)
    b.set_state(torch.from_numpy(np.stack(q0s)), torch.from_numpy(np.stack(v0s)))
    act = torch.zeros(n, b.ctrl.action_dim, dtype=torch.float64, device="cuda")
    swept = torch.zeros(n, dtype=torch.int64, device="cuda")
    for t in range(settle + n_steps):
        if applied is not None and t == settle:
            b.field(S.F_QFRC_APPLIED).copy_(torch.from_numpy(np.stack(applied)).cuda())
        b.simulate(act, act)
        swept += ((b.field(S.F_REDO) & 2) != 0).long()  # steps whose exact solve gave up and swept to tolerance (general tier)
    b.sync()
    b.swept_steps = swept.cpu().numpy()
    return b


def _oracle(model, q0, v0, n_steps, applied=None, settle=0, solver=None):
    from oracle.physics import OracleSim
    s = OracleSim(model)
    if solver is not None:
        s.desc.solver = solver
    s.set_state(q0, v0)
    for t in range(settle + n_steps):
        if applied is not None and t == settle:
            s.set("qfrc_applied", applied)
        s.step()
    return s


def test_friction_cone_slip_threshold_gpu(kernel_path):
    from uhc_amd import sim as S
    tans = [0.8, 0.95, 1.1, 1.5]
    ms, q0s = zip(*[scenario_tilted_gravity_box(t) for t in tans])
    n = 500
    q = np.zeros((4, 7))
    swept = np.zeros(4, dtype=int)
    for e in range(4):  # gravity is a batch-wide option (MuJoCo's opt.gravity): one batch per slope
        b = _run([ms[e]], [q0s[e]], [np.zeros(6)], n)
        q[e] = b.field(S.F_QPOS).cpu().numpy()[0]
        swept[e] = b.swept_steps[0]
        assert b.field(S.F_FAIL).sum().item() == 0
    t = n * ms[0].timestep
    for e, tt in enumerate(tans):
        th = np.arctan(tt)
        if tt > 1:
            assert q[e, 0] == pytest.approx(0.5 * 9.81 * (np.sin(th) - np.cos(th)) * t * t, rel=0.12)
        else:
            assert abs(q[e, 0]) < 0.02 * 0.5 * 9.81 * np.sin(th) * t * t
        # and the oracle's trajectory.  Both tiers solve the QP exactly (active set in registers / working sets) and then agree with the
        # oracle's exact solve to rounding over the 500 steps; only where the general tier's working sets gave up in some step and swept to
        # the PGS tolerance (UHC_F_REDO bit 1, counted per env) the free-running oracle -- exact in every step -- is followed to 5e-5 only
        o = _oracle(ms[e], q0s[e], np.zeros(6), n, solver=1)
        print(f"tan(theta) = {tt}: steps solved by sweeps {swept[e]} / {n}, |gpu - oracle| = {np.abs(q[e] - o.get('qpos')).max():.2e}")
        np.testing.assert_allclose(q[e], o.get("qpos"), atol=5e-5 if swept[e] else 1e-7)


def test_tipping_threshold_gpu(kernel_path):
    from uhc_amd import sim as S
    cases = [scenario_pushed_box(0.8), scenario_pushed_box(1.2)]
    b = _run([cases[0][0]], [c[1] for c in cases], [np.zeros(6)] * 2, 400, applied=[c[2] for c in cases], settle=100)
    q = b.field(S.F_QPOS).cpu().numpy()
    tilt = 2 * np.arccos(np.minimum(1.0, np.abs(q[:, 3])))
    assert tilt[0] < 0.02 and tilt[1] > 0.5


def test_resting_equilibrium_and_flat_contacts_gpu(kernel_path):
    """The prism comes to rest where the oracle's does (whose penetration the closed-form solref / solimp balance pins), on three
    contacts; the flat box reports the oracle's contact count."""
    import torch
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import compile_mjcf
    from tests.helpers import BOX_ON_PLANE_XML, box_model, prism_triangles
    m = compile_mjcf(BOX_ON_PLANE_XML, meshes={"box": prism_triangles(0.1, 0.05)})
    m.solver = 1
    q0 = np.array([0, 0, 0.0499, 1, 0, 0, 0.0])
    b = _run([m], [q0], [np.zeros(6)], 1500)
    o = _oracle(m, q0, np.zeros(6), 1500, solver=1)
    np.testing.assert_allclose(b.field(S.F_QPOS).cpu().numpy()[0], o.get("qpos"), atol=1e-6)
    assert int(b.field(S.F_NCON)[0].item()) == 3 and abs(b.field(S.F_QVEL).cpu().numpy()).max() < 1e-5
    mb = box_model(0.1)
    sb = S.SimBatch(mb, passive_ctrl(mb), 1)
    qb = np.array([[0.3, -0.2, 0.1004, 1, 0, 0, 0.0]])
    sb.set_state(torch.from_numpy(qb), torch.zeros(1, 6, dtype=torch.float64))
    sb.sync()
    from oracle.physics import OracleSim
    ob = OracleSim(mb)
    ob.set_state(qb[0], np.zeros(6))
    assert int(sb.field(S.F_NCON)[0].item()) == ob.geti("ncon") and int(sb.field(S.F_NEFC)[0].item()) == ob.geti("nefc")
    np.testing.assert_allclose(sb.field(S.F_QACC).cpu().numpy()[0], ob.get("qacc"), atol=1e-6)


def test_implicit_joint_damping_gpu(kernel_path):
    """mj_Euler's implicit damping on the device: the closed-form backward-Euler decay of a damped hinge, and a damped humanoid
    (every hinge damped, as the copycat_ball configs set) against the oracle over 20 control steps."""
    import dataclasses
    import torch
    from oracle.physics import OracleSim
    from tests.helpers import pendulum_model
    from uhc_amd import sim as S
    m = pendulum_model(length=0.5, half=0.05)
    m.gravity = np.zeros(3)
    m.dof_damping = np.array([0.7])
    I = 1.0 / m.dof_invweight0[0]
    b = S.SimBatch(m, passive_ctrl(m), 2)
    b.set_state(torch.tensor([[0.1], [0.2]], dtype=torch.float64), torch.tensor([[3.0], [-1.0]], dtype=torch.float64))
    act = torch.zeros(2, 1, dtype=torch.float64, device="cuda")
    for _ in range(400):
        b.simulate(act, act)
    b.sync()
    np.testing.assert_allclose(b.field(S.F_QVEL).cpu().numpy()[:, 0], np.array([3.0, -1.0]) * (I / (I + m.timestep * 0.7)) ** 400, rtol=1e-10)

    from uhc_amd.sim import load_asset_model, make_ctrl
    hm = load_asset_model()
    hm = dataclasses.replace(hm, solver=1, dof_damping=np.r_[np.zeros(6), np.full(69, 5.0)])
    ctrl = make_ctrl(hm)
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uhc_amd", "assets", "standing_neutral.npz"))
    rng = np.random.default_rng(21)
    n = 3
    qpos = np.tile(z["qpos"], (n, 1))
    qpos[:, 7:] += rng.normal(scale=0.05, size=(n, 69))
    qvel = rng.normal(scale=0.3, size=(n, 75))
    hb = S.SimBatch(hm, ctrl, n)
    hb.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
    os_ = [OracleSim(hm, ctrl) for _ in range(n)]
    for e in range(n):
        os_[e].set_state(qpos[e], qvel[e])
    for t in range(20):
        act = rng.normal(scale=0.05, size=(n, ctrl.action_dim))
        hb.simulate(torch.from_numpy(act).cuda(), tb)
        hb.sync()
        redo = hb.field(S.F_REDO).cpu().numpy()
        for e in range(n):
            os_[e].do_simulation(act[e], qpos[e, 7:], redo=redo[e])  # UHC_F_REDO bits 8+: the substeps the general kernel solved by sweeps
    gq = hb.field(S.F_QPOS).cpu().numpy()
    for e in range(n):
        np.testing.assert_allclose(gq[e], os_[e].get("qpos"), atol=1e-6)
    # damping matters at all: the undamped model ends elsewhere
    o2 = OracleSim(dataclasses.replace(hm, dof_damping=np.zeros(75)), ctrl)
    o2.set_state(qpos[0], qvel[0])
    o2.do_simulation(np.zeros(ctrl.action_dim), qpos[0, 7:])
    o3 = OracleSim(hm, ctrl)
    o3.set_state(qpos[0], qvel[0])
    o3.do_simulation(np.zeros(ctrl.action_dim), qpos[0, 7:])
    assert np.abs(o2.get("qpos") - o3.get("qpos")).max() > 1e-4
