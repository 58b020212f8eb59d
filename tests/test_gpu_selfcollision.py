"""GPU parity for the convex-convex narrow phase (MPR, one pair per lane) and for contact rows between two moving bodies (dense rows):
free boxes (two kinematic trees) and the humanoid with body-body collisions on, as the reference's generated models have them
(uhc/smpllib/smpl_parser.py:327-328, uhc/smpllib/smpl_robot.py:1177-1198).  HIP path through the C-ABI vs the CPU oracle."""
import dataclasses
import os

import numpy as np
import pytest

from tests.helpers import passive_ctrl, two_box_model

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


def _box_poses(n, seed):
    """n poses of the two boxes high above the floor: box B near a face / edge / corner of box A, random orientations."""
    from scipy.spatial.transform import Rotation as sR
    rng = np.random.default_rng(seed)
    q = np.zeros((n, 14))
    for e in range(n):
        qa = sR.from_rotvec(rng.normal(size=3) * 0.5)
        qb = sR.from_rotvec(rng.normal(size=3) * 0.9)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        q[e, :3] = [0, 0, 2.0]
        q[e, 3:7] = np.roll(qa.as_quat(), 1)
        q[e, 7:10] = q[e, :3] + d * rng.uniform(0.12, 0.2)
        q[e, 10:14] = np.roll(qb.as_quat(), 1)
    return q


def test_mpr_contacts_match_oracle(kernel_path):
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    m = dataclasses.replace(two_box_model(0.1, 0.06), solver=1)
    n = 64
    q = _box_poses(n, 3)
    v = np.random.default_rng(4).normal(scale=0.3, size=(n, 12))
    b = S.SimBatch(m, passive_ctrl(m), n)
    b.set_state(torch.from_numpy(q), torch.from_numpy(v))
    b.sync()
    ncon, nefc, qacc = (b.field(f).cpu().numpy() for f in (S.F_NCON, S.F_NEFC, S.F_QACC))
    hits = 0
    for e in range(n):
        o = OracleSim(m)
        o.desc.solver = 0 if (int(b.field(S.F_REDO)[e].item()) & 2) else 1
        o.set_state(q[e], v[e])
        assert ncon[e] == o.geti("ncon") and nefc[e] == o.geti("nefc"), e
        np.testing.assert_allclose(qacc[e], o.get("qacc"), atol=1e-6, rtol=1e-6)
        hits += o.geti("ncon")
    assert 10 < hits < n  # the sample has touching and separated pairs


def test_stacked_boxes_trajectory_matches_oracle(kernel_path):
    """Two trees, floor contacts (chain rows) + a box-box contact (dense row) in one solve, 120 steps."""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    m = dataclasses.replace(two_box_model(0.1, 0.06), solver=1)
    q0 = np.array([[0, 0, 0.0998, 1, 0, 0, 0, 0.01, -0.02, 0.2 + 0.0598, 1, 0, 0, 0.0],
                   [0, 0, 0.0998, 1, 0, 0, 0, 0.05, 0.03, 0.2 + 0.07, 0.9689, 0.2474, 0, 0.0]])
    b = S.SimBatch(m, passive_ctrl(m), 2)
    b.set_state(torch.from_numpy(q0), torch.zeros(2, 12, dtype=torch.float64))
    act = torch.zeros(2, 1, dtype=torch.float64, device="cuda")
    os_ = [OracleSim(m) for _ in range(2)]
    for e in range(2):
        os_[e].desc.solver = 1
        os_[e].set_state(q0[e], np.zeros(12))
    worst = 0.0
    for t in range(120):
        b.simulate(act, act)
        b.sync()
        gq = b.field(S.F_QPOS).cpu().numpy()
        for e in range(2):
            os_[e].step()
            worst = max(worst, np.abs(gq[e] - os_[e].get("qpos")).max())
    assert worst < 1e-6, worst
    assert int(b.field(S.F_FAIL).sum().item()) == 0 and gq[0, 9] > 0.24  # still stacked


def _humanoid_states(standing, n, seed, lift):
    rng = np.random.default_rng(seed)
    qpos = np.tile(standing["qpos"], (n, 1))
    qpos[:, 7:] += rng.normal(scale=0.08, size=(n, 69))
    qpos[:, 2] += lift
    return qpos, rng.normal(scale=0.3, size=(n, 75))


@pytest.mark.parametrize("lift", [1.0, 0.0])
def test_self_collision_humanoid_matches_oracle(model, standing, kernel_path, lift):
    """Body-body contacts of the humanoid: forward fields and a 20-control-step trajectory.  lift = 1: airborne, only self-contacts
    (the fast kernel's dense-row path); lift = 0: standing, floor + self contacts (> 64 rows: redone by the general kernel)."""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import self_collision_variant
    from uhc_amd.sim import make_ctrl
    sc = dataclasses.replace(self_collision_variant(model), solver=1)
    ctrl = make_ctrl(sc)
    n = 6
    qpos, qvel = _humanoid_states(standing, n, 31, lift)
    b = S.SimBatch(sc, ctrl, n)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    b.sync()
    os_ = [OracleSim(sc, ctrl) for _ in range(n)]
    ncon, nefc, qacc, redo = (b.field(f).cpu().numpy() for f in (S.F_NCON, S.F_NEFC, S.F_QACC, S.F_REDO))
    nself = 0
    for e in range(n):
        os_[e].desc.solver = 0 if (redo[e] & 2) else 1
        os_[e].set_state(qpos[e], qvel[e])
        assert ncon[e] == os_[e].geti("ncon") and nefc[e] == os_[e].geti("nefc"), (e, ncon[e], os_[e].geti("ncon"))
        np.testing.assert_allclose(qacc[e], os_[e].get("qacc"), atol=1e-5, rtol=1e-6)
        nself += os_[e].geti("ncon")
    assert nself > 0
    if lift > 0 and kernel_path == "fast":
        assert (redo != 0).sum() <= 1  # few rows: the fast kernel's dense path did the work (an env with > 12 body-body rows goes to the general kernel)
    for o in os_:
        o.desc.solver = 1  # (the forward pass of set_state may have been compared under solver 0, see above)
    rng = np.random.default_rng(32)
    tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
    worst = 0.0
    for t in range(20):
        act = rng.normal(scale=0.05, size=(n, ctrl.action_dim))
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        gq = b.field(S.F_QPOS).cpu().numpy()
        redo = b.field(S.F_REDO).cpu().numpy()
        for e in range(n):
            os_[e].do_simulation(act[e], qpos[e, 7:], redo=redo[e])  # UHC_F_REDO bits 8+: the substeps the general kernel solved by sweeps
            worst = max(worst, np.abs(gq[e] - os_[e].get("qpos")).max())
    assert worst < 1e-5, worst
    assert int(b.field(S.F_EFC_OVERFLOW).sum().item()) == 0


def test_sticky_tiers_follow_the_scene(model, standing):
    """uhc_batch_set_kernel_path(2): seven self-colliding humanoids standing on the floor (> 64 rows each) and one in the air.  In the
    first step every env starts in the fast tier, which hands the standing ones on (UHC_F_TIER becomes 2); from the second step on they
    start in the general tier, on the side stream beside the fast tier's launch, while the airborne env stays with the fast tier
    (UHC_F_REDO 0).  With everybody lifted into the air the general tier computes its envs once more and lets them go (room to spare):
    the next step is all fast tier.  The states follow the oracle through it all, and equal a run on the plain tier chain up to the
    rounding of the two exact solvers.  (Every step restarts from the same states, so the scene stays what it is.)"""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import self_collision_variant
    from uhc_amd.sim import make_ctrl
    if os.environ.get("UHC_FORCE_GENERAL") == "1":
        pytest.skip("the batch has no fast tier to start from")
    sc = dataclasses.replace(self_collision_variant(model), solver=1)
    ctrl = make_ctrl(sc)
    n = 8
    rng = np.random.default_rng(61)
    qpos = np.tile(standing["qpos"], (n, 1))
    qpos[:, 7:] += rng.normal(scale=0.002, size=(n, 69))
    qpos[7, 2] += 50.0
    qvel = rng.normal(scale=0.01, size=(n, 75))
    b = S.SimBatch(sc, ctrl, n)
    b.set_kernel_path(2)
    os_ = [OracleSim(sc, ctrl) for _ in range(n)]
    tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
    act = np.zeros((n, ctrl.action_dim))
    a = torch.from_numpy(act).cuda()

    def steps(q, k):
        worst, hist, tiers = 0.0, [], []
        for _ in range(k):
            b.set_state(torch.from_numpy(q), torch.from_numpy(qvel))
            b.simulate(a, tb)
            b.sync()
            redo, gq = b.field(S.F_REDO).cpu().numpy().copy(), b.field(S.F_QPOS).cpu().numpy()
            hist.append(redo)
            tiers.append(b.field(S.F_TIER).cpu().numpy().copy())
            for e in range(n):
                os_[e].set_state(q[e], qvel[e])
                os_[e].do_simulation(act[e], qpos[e, 7:], redo=redo[e])
                worst = max(worst, np.abs(gq[e] - os_[e].get("qpos")).max())
        return worst, np.array(hist), np.array(tiers), gq

    assert (b.field(S.F_TIER).cpu().numpy() == 1).all()
    w1, h1, t1, q_sticky = steps(qpos, 4)
    heavy = h1[0, :7] != 0
    assert heavy.sum() >= 6 and (h1[:, 7] == 0).all()  # the standing envs are beyond the fast tier, the airborne one never is
    assert (t1[:, :7][:, heavy] == 2).all() and (t1[:, 7] == 1).all()  # ... and they stay with the general tier
    assert (h1[1:, :7][:, heavy] != 0).all()
    chain = S.SimBatch(sc, ctrl, n)  # the same step on the plain chain: same physics
    chain.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    chain.simulate(a, tb)
    chain.sync()
    np.testing.assert_allclose(chain.field(S.F_QPOS).cpu().numpy(), q_sticky, atol=1e-10)
    lifted = qpos.copy()
    lifted[:, 2] += 50.0
    w2, h2, t2, _ = steps(lifted, 3)
    assert (h2[0, :7][heavy] != 0).all() and h2[0, 7] == 0  # airborne now, but this step still starts where the last one ended
    assert (t2[0] == 1).all() and (h2[1:] == 0).all()       # ... came down with room to spare: fast tier from the next step on
    assert max(w1, w2) < 1e-9, (w1, w2)


def test_hand_on_resumes_at_the_substep(model, standing):
    """A tier that finds an env too big in the middle of a control step hands it on WITH the substeps it has done: the next tier goes on
    from the substep that did not fit.  Self-colliding humanoids dropped from 2-7 cm: the feet land during the third or fourth control step, and the
    impact takes the rows past 64 in one of its substeps.  The tier trace (UHC_DEBUG bit 4) shows hand-ons at a substep >= 1; the states equal those of a build
    switch that makes the next tier repeat the step (bit 2) up to the rounding of the tiers' solvers, and follow the oracle."""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import self_collision_variant
    from uhc_amd.sim import make_ctrl
    if os.environ.get("UHC_FORCE_GENERAL") == "1":
        pytest.skip("the batch has no fast tier to hand on from")
    sc = dataclasses.replace(self_collision_variant(model), solver=1)
    ctrl = make_ctrl(sc)
    n = 6
    rng = np.random.default_rng(67)
    qpos = np.tile(standing["qpos"], (n, 1))
    qpos[:, 7:] += rng.normal(scale=0.002, size=(n, 69))
    qpos[:, 2] += np.linspace(0.02, 0.07, n)
    qvel = np.zeros((n, 75))
    old = os.environ.get("UHC_DEBUG")
    try:
        os.environ["UHC_DEBUG"] = "16"
        resume = S.SimBatch(sc, ctrl, n)
        os.environ["UHC_DEBUG"] = "20"
        restart = S.SimBatch(sc, ctrl, n)
    finally:
        if old is None:
            os.environ.pop("UHC_DEBUG", None)
        else:
            os.environ["UHC_DEBUG"] = old
    os_ = [OracleSim(sc, ctrl) for _ in range(n)]
    tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
    act = np.zeros((n, ctrl.action_dim))
    a = torch.from_numpy(act).cuda()
    for bb in (resume, restart):
        bb.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    for e in range(n):
        os_[e].set_state(qpos[e], qvel[e])
    substeps, worst, apart = [], 0.0, 0.0
    for _ in range(6):
        resume.field(S.F_STAGE_PROF).zero_()
        for bb in (resume, restart):
            bb.simulate(a, tb)
            bb.sync()
        redo, gq = resume.field(S.F_REDO).cpu().numpy(), resume.field(S.F_QPOS).cpu().numpy()
        tr = resume.field(S.F_STAGE_PROF).cpu().numpy()
        substeps.append(np.where((redo & 1) != 0, tr[:, 6], -1))  # word 6 of the trace: the substep the fast tier handed the env on in
        apart = max(apart, np.abs(gq - restart.field(S.F_QPOS).cpu().numpy()).max())
        for e in range(n):
            os_[e].do_simulation(act[e], qpos[e, 7:], redo=redo[e])
            worst = max(worst, np.abs(gq[e] - os_[e].get("qpos")).max())
    substeps = np.array(substeps)
    assert (substeps >= 1).any(), substeps  # some env was handed on in mid-step ...
    assert apart < 1e-8 and worst < 1e-5, (apart, worst)
    assert int(resume.field(S.F_EFC_OVERFLOW).sum().item()) == 0


def test_lying_humanoid_among_boxes_exceeds_128_rows(model, standing):
    """The reference's models ask MuJoCo for njmax 2500 / nconmax 500 (uhc/khrylib/mocap/skeleton_mesh.py:46).  A self-colliding
    humanoid lying face down among four boxes has 100-150 constraint rows: beyond the general tier (128 rows / 64 contacts), so the env is
    handed to the large tier (256 rows, UHC_F_REDO bit 6) instead of losing constraints.  GPU vs oracle over 12 control steps, and
    nothing is dropped (UHC_F_EFC_OVERFLOW stays clear)."""
    import dataclasses
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import add_free_bodies, quat_mul, self_collision_variant
    from uhc_amd.model.shapes import box_triangles
    from uhc_amd.sim import make_ctrl
    m = self_collision_variant(model)
    poses = np.array([[0.6, 0.3, 0.16, 1, 0, 0, 0], [-0.6, 0.2, 0.16, 1, 0, 0, 0], [0.1, 0.9, 0.16, 1, 0, 0, 0], [0.0, -0.7, 0.16, 1, 0, 0, 0]], dtype=np.float64)
    m = add_free_bodies(m, [box_triangles(0.15, 0.15, 0.15)] * 4, poses, density=5.0 / 0.027)
    m = dataclasses.replace(m, solver=1)
    ctrl = make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
    n = 3
    rng = np.random.default_rng(3)
    q = np.tile(m.qpos0, (n, 1))
    for e in range(n):
        qh = standing["qpos"].copy()
        qh[7:] += rng.normal(scale=0.05 * e, size=69)
        a = np.pi / 2 + 0.1 * e
        qh[3:7] = quat_mul(np.array([np.cos(a / 2), 0, np.sin(a / 2), 0]), qh[3:7])  # tipped forward: face down
        qh[2] = 0.25
        q[e, :76] = qh
    v = np.zeros((n, m.nv))
    b = S.SimBatch(m, ctrl, n)
    b.set_state(torch.from_numpy(q), torch.from_numpy(v))
    b.sync()
    os_ = [OracleSim(m, ctrl) for _ in range(n)]
    redo = b.field(S.F_REDO).cpu().numpy()
    for e in range(n):
        os_[e].desc.solver = 0 if (redo[e] & 2) else 1
        os_[e].set_state(q[e], v[e])
        os_[e].desc.solver = 1
    tb = torch.zeros(n, 69, dtype=torch.float64, device="cuda")
    worst, big_steps, max_nefc = 0.0, 0, 0
    for t in range(12):
        act = rng.normal(scale=0.003, size=(n, ctrl.action_dim))
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        gq = b.field(S.F_QPOS).cpu().numpy()
        redo = b.field(S.F_REDO).cpu().numpy()
        nefc = b.field(S.F_NEFC).cpu().numpy()
        for e in range(n):
            os_[e].do_simulation(act[e], np.zeros(69), redo=redo[e])
            assert nefc[e] == os_[e].geti("nefc"), (t, e, nefc[e], os_[e].geti("nefc"))
            worst = max(worst, np.abs(gq[e] - os_[e].get("qpos")).max())
            big_steps += int((redo[e] & 0x40) != 0)
            max_nefc = max(max_nefc, os_[e].geti("max_nefc"))
    print(f"lying humanoid + 4 boxes: max nefc {max_nefc}, env-steps in the large tier {big_steps} / {12 * n}, worst |dqpos| {worst:.2e}")
    assert max_nefc > 128 and big_steps > 0
    assert int(b.field(S.F_EFC_OVERFLOW).sum().item()) == 0 and int(b.field(S.F_FAIL).sum().item()) == 0
    assert worst < 1e-5, worst


@pytest.mark.parametrize("tiers", ["4", "3"])
def test_island_with_more_than_64_force_rows_is_solved_exactly(model, standing, tiers, monkeypatch):
    """(tiers = 4, the default: the general / large tier hands such an island on to tier 4, whose Newton iteration on the primal problem -- MuJoCo's own
    default solver -- has no limit on the rows that carry a force: UHC_F_REDO bit 30, no windows, no sweeps.  tiers = 3, UHC_TIERS=3: the three-tier chain of
    rounds 3-4, described next.)
    MuJoCo's Newton solver (humanoid_template.xml:13) solves the contact QP to 1e-8 whatever its size.  Here the exact solve keeps the
    Delassus matrix of at most 64 rows in registers; an island with more force-carrying rows than that is solved in windows of 64 rows
    (block coordinate descent to a KKT residual of 1e-9 (1 + max |b|), UHC_F_REDO bit 3) instead of falling back to the sweeps (bit 1).
    Scene: seven 5 kg boxes side by side on the floor, yawed +-3.4 degrees so that every corner digs a millimetre into its neighbour --
    one island of 90-120 rows (28 floor contacts and 6 box-box contacts, pyramids of 4), 61-78 of them with a force in the first four
    control steps (later nearly all of them carry one and the windows need more rounds than they are given: tools/proto_block_cd.py)
    -- beside a humanoid.  Every control step is checked against the oracle, which solves all rows at once, started from the
    device's state (the scene is a pile-up: trajectories of two solvers that agree to 1e-9 per step still drift apart)."""
    import dataclasses
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import add_free_bodies, self_collision_variant
    from uhc_amd.model.shapes import box_triangles
    from uhc_amd.sim import make_ctrl
    monkeypatch.setenv("UHC_TIERS", tiers)
    K = 7
    m = self_collision_variant(model)
    yaw = [0.06 * (-1) ** k for k in range(K)]
    poses = np.array([[1.0 + 0.305 * k, 1.0 + 0.01 * k, 0.1495, np.cos(y / 2), 0, 0, np.sin(y / 2)] for k, y in enumerate(yaw)], dtype=np.float64)
    m = add_free_bodies(m, [box_triangles(0.15, 0.15, 0.15)] * K, poses, density=5.0 / 0.027)
    m = dataclasses.replace(m, solver=1)
    ctrl = make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
    n = 2
    q = np.tile(m.qpos0, (n, 1))
    q[:, :76] = standing["qpos"]
    q[1, 76 + 7 * 3] += 0.0005  # (env 1: the fourth box half a millimetre further along)
    v = np.zeros((n, m.nv))
    b = S.SimBatch(m, ctrl, n)
    b.set_state(torch.from_numpy(q), torch.from_numpy(v))
    b.sync()
    os_ = [OracleSim(m, ctrl) for _ in range(n)]
    tb = torch.zeros(n, 69, dtype=torch.float64, device="cuda")
    act = np.zeros((n, ctrl.action_dim))
    windowed, swept, primal, worst_q, worst_v, max_nefc = 0, 0, 0, 0.0, 0.0, 0
    for t in range(4):
        gq0, gv0 = b.field(S.F_QPOS).cpu().numpy().copy(), b.field(S.F_QVEL).cpu().numpy().copy()
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        gq, gv = b.field(S.F_QPOS).cpu().numpy(), b.field(S.F_QVEL).cpu().numpy()
        redo = b.field(S.F_REDO).cpu().numpy()
        for e in range(n):
            windowed += int((redo[e] & 8) != 0)
            primal += int((redo[e] & (1 << 30)) != 0)
            assert not (redo[e] & (1 << 29)), hex(int(redo[e]))  # (the Newton iteration never ran into its cap)
            swept += int((redo[e] & 2) != 0)
            os_[e].set_state(gq0[e], gv0[e])
            os_[e].do_simulation(act[e], np.zeros(69))
            dq, dv = np.abs(gq[e] - os_[e].get("qpos")).max(), np.abs(gv[e] - os_[e].get("qvel")).max()
            worst_q, worst_v = max(worst_q, dq), max(worst_v, dv)
            max_nefc = max(max_nefc, os_[e].geti("max_nefc"))
    print(f"raft of {K} boxes, tiers {tiers}: env-steps with a windowed exact solve {windowed} / {4 * n}, solved by Newton on the primal {primal}, sweeps fallbacks {swept}, max nefc {max_nefc}; "
          f"one control step from the device's state, device vs oracle: |dqpos| {worst_q:.2e} |dqvel| {worst_v:.2e}")
    if tiers == "3":
        assert windowed > 0 and primal == 0, "no island exceeded 64 force-carrying rows: the scene no longer exercises the windows"
    else:
        assert primal > 0 and windowed == 0, (primal, windowed)
    assert swept == 0
    assert int(b.field(S.F_EFC_OVERFLOW).sum().item()) == 0 and int(b.field(S.F_FAIL).sum().item()) == 0
    assert worst_q < 1e-10 and worst_v < 1e-8, (worst_q, worst_v)


def test_pass_that_drops_rows_is_flagged_and_bounded(model, standing, monkeypatch):
    """(UHC_TIERS=3: the three-tier chain.  With tier 4 behind the large tier -- the default -- these rows are not dropped: next test.)
    Beyond the large tier's 256 rows constraint rows are dropped (the reference's njmax is 2500: uhc/khrylib/mocap/skeleton_mesh.py:46) and the
    pass is no longer the reference's QP.  It is reported where it happens -- UHC_F_REDO bit 7 of that step, UHC_F_EFC_OVERFLOW until the next
    set_state -- and its truncated QP gets a bounded exact attempt (six working-set rounds, no windows) and, if that gives up, at most 32
    sweeps (bit 1 with bit 7, and none of the "gave up" reasons that would mean 300 sweeps).  Scene: a
    humanoid laid flat into the floor (its chest 16 cm up: 130-190 rows) beside the seven-box raft of the test above (110-120 rows): 270-330
    rows in the first step."""
    import dataclasses
    import torch
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import add_free_bodies, quat_mul, self_collision_variant
    from uhc_amd.model.shapes import box_triangles
    from uhc_amd.sim import make_ctrl
    monkeypatch.setenv("UHC_TIERS", "3")
    K = 7
    m = self_collision_variant(model)
    yaw = [0.06 * (-1) ** k for k in range(K)]
    poses = np.array([[1.0 + 0.305 * k, 1.0 + 0.01 * k, 0.1495, np.cos(y / 2), 0, 0, np.sin(y / 2)] for k, y in enumerate(yaw)], dtype=np.float64)
    m = add_free_bodies(m, [box_triangles(0.15, 0.15, 0.15)] * K, poses, density=5.0 / 0.027)
    m = dataclasses.replace(m, solver=1)
    ctrl = make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
    n = 2
    q = np.tile(m.qpos0, (n, 1))
    for e in range(n):
        qh = standing["qpos"].copy()
        a = np.pi / 2 + 0.1 * e
        qh[3:7] = quat_mul(np.array([np.cos(a / 2), 0, np.sin(a / 2), 0]), qh[3:7])  # tipped forward: face down
        qh[0], qh[1], qh[2] = -0.8, -0.8, 0.14
        q[e, :76] = qh
    v = np.zeros((n, m.nv))
    b = S.SimBatch(m, ctrl, n)
    b.set_state(torch.from_numpy(q), torch.from_numpy(v))
    tb = torch.zeros(n, 69, dtype=torch.float64, device="cuda")
    act = torch.zeros(n, ctrl.action_dim, dtype=torch.float64, device="cuda")
    lost_steps, nefc_max, words, swept = 0, 0, [], 0
    for t in range(4):
        b.simulate(act, tb)
        b.sync()
        redo = b.field(S.F_REDO).cpu().numpy()
        nefc_max = max(nefc_max, int(b.field(S.F_NEFC).max().item()))
        for e in range(n):
            if redo[e] & 0x80:
                lost_steps += 1
                words.append(hex(int(redo[e])))
                assert redo[e] & 0x40, hex(int(redo[e]))  # in the large tier
                swept += int((redo[e] & 2) != 0)
    print(f"humanoid face down + raft of {K} boxes: {lost_steps} of {4 * n} env-steps lost rows beyond 256 (nefc at the steps' ends up to {nefc_max}), {swept} of them ended a substep in the short sweeps; UHC_F_REDO of those: {words[:4]}")
    assert lost_steps > 0  # (UHC_F_NEFC is the step's LAST substep: the rows were lost in its first ones)
    assert int(b.field(S.F_EFC_OVERFLOW).sum().item()) > 0 and int(b.field(S.F_FAIL).sum().item()) == 0
    assert np.isfinite(b.field(S.F_QPOS).cpu().numpy()).all()


@pytest.mark.parametrize("sticky4", [False, True], ids=["default", "sticky_tier4"])
def test_more_rows_than_the_dual_tiers_hold_are_solved_by_tier_4(model, standing, sticky4, monkeypatch):
    """The reference asks MuJoCo for njmax 2500 / nconmax 500 (uhc/khrylib/mocap/skeleton_mesh.py:46) and solves with Newton on the primal.  The scene of
    the test above -- a humanoid face down in the floor beside the seven-box raft, 270-330 rows in the first substeps, more than 150 of them
    carrying a force -- is beyond the 256 rows / 128 contacts of the large tier: its workgroup goes on as tier 4 (rows in HBM, the nv x nv
    Hessian in LDS, uhc_primal.h).  Nothing is dropped (UHC_F_EFC_OVERFLOW clear, UHC_F_REDO bit 7 clear), bit 30 reports the primal solve, and
    every control step from the device's own state equals the oracle's (which takes its own primal path above 256 rows) to 1e-9."""
    import dataclasses
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import add_free_bodies, quat_mul, self_collision_variant
    from uhc_amd.model.shapes import box_triangles
    from uhc_amd.sim import make_ctrl
    if sticky4:
        monkeypatch.setenv("UHC_T4_ROWS", "200")  # (an env whose step peaked at 200 rows or more starts its next step in tier 4, at the head of its consumers' queue -- KernelArgs::t4_rows)
    K = 7
    m = self_collision_variant(model)
    yaw = [0.06 * (-1) ** k for k in range(K)]
    poses = np.array([[1.0 + 0.305 * k, 1.0 + 0.01 * k, 0.1495, np.cos(y / 2), 0, 0, np.sin(y / 2)] for k, y in enumerate(yaw)], dtype=np.float64)
    m = add_free_bodies(m, [box_triangles(0.15, 0.15, 0.15)] * K, poses, density=5.0 / 0.027)
    m = dataclasses.replace(m, solver=1)
    ctrl = make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
    n = 3
    q = np.tile(m.qpos0, (n, 1))
    for e in range(n):
        qh = standing["qpos"].copy()
        a = np.pi / 2 + 0.1 * e
        qh[3:7] = quat_mul(np.array([np.cos(a / 2), 0, np.sin(a / 2), 0]), qh[3:7])  # tipped forward: face down
        qh[0], qh[1], qh[2] = -0.8, -0.8, 0.14
        q[e, :76] = qh
    v = np.zeros((n, m.nv))
    for mode in (0, 2):  # the tier chain, and the sticky queues (the large tier's consumers go on as tier 4)
        b = S.SimBatch(m, ctrl, n)
        b.set_kernel_path(mode)
        b.set_state(torch.from_numpy(q), torch.from_numpy(v))
        b.sync()
        os_ = [OracleSim(m, ctrl) for _ in range(n)]
        for e in range(n):  # the forward pass of set_state: already more rows than the large tier holds
            os_[e].set_state(q[e], v[e])
            assert os_[e].geti("nefc") > 256
            assert int(b.field(S.F_NEFC)[e].item()) == os_[e].geti("nefc") and int(b.field(S.F_NCON)[e].item()) == os_[e].geti("ncon")
            np.testing.assert_allclose(b.field(S.F_QACC)[e].cpu().numpy(), os_[e].get("qacc"), atol=1e-7 * (1 + np.abs(os_[e].get("qacc")).max()))
        tb = torch.zeros(n, 69, dtype=torch.float64, device="cuda")
        act = np.zeros((n, ctrl.action_dim))
        primal, worst_q, worst_v, max_nefc = 0, 0.0, 0.0, 0
        for t in range(8):
            gq0, gv0 = b.field(S.F_QPOS).cpu().numpy().copy(), b.field(S.F_QVEL).cpu().numpy().copy()
            b.simulate(torch.from_numpy(act).cuda(), tb)
            b.sync()
            gq, gv = b.field(S.F_QPOS).cpu().numpy(), b.field(S.F_QVEL).cpu().numpy()
            redo = b.field(S.F_REDO).cpu().numpy()
            for e in range(n):
                assert not (redo[e] & 0x80) and not (redo[e] & 2) and not (redo[e] & (1 << 29)), (t, e, hex(int(redo[e])))
                primal += int((redo[e] & (1 << 30)) != 0)
                os_[e].set_state(gq0[e], gv0[e])
                os_[e].do_simulation(act[e], np.zeros(69))
                worst_q = max(worst_q, np.abs(gq[e] - os_[e].get("qpos")).max())
                worst_v = max(worst_v, np.abs(gv[e] - os_[e].get("qvel")).max())
                max_nefc = max(max_nefc, os_[e].geti("max_nefc"))
        print(f"face-down humanoid + raft of {K} boxes, kernel path {mode}: up to {max_nefc} rows, {primal} of {8 * n} env-steps went through tier 4; one control step from the "
              f"device's state, device vs oracle: |dqpos| {worst_q:.2e} |dqvel| {worst_v:.2e}")
        assert max_nefc > 256 and primal > 0
        assert int(b.field(S.F_EFC_OVERFLOW).sum().item()) == 0 and int(b.field(S.F_FAIL).sum().item()) == 0
        assert worst_q < 1e-9 and worst_v < 1e-7, (worst_q, worst_v)
        if mode == 2 and sticky4:  # opt-in: an env that needed tier 4 starts its next step there (its own launch from the head of the step once the host has seen the count)
            tiers, rows = b.field(S.F_TIER).cpu().numpy(), b.field(S.F_NEFC).cpu().numpy()
            assert (tiers[rows >= 200] == 4).all() and (tiers == 4).any() and (tiers >= 3).all(), (tiers.tolist(), rows.tolist())
        b.close()


def test_sticky_queues_hand_an_env_on_to_tier_4(model, standing):
    """Kernel path 2 with consumers running: a batch whose envs live in different tiers -- some standing (general tier: the queue's own envs, so that the
    consumers are launched), one in the air (fast tier), two face down beside the raft (beyond the large tier: handed from queue to queue and on to tier
    4 INSIDE a large-tier consumer).  Several steps, every env re-posed each step so that the scene stays what it is; against the oracle, step by step."""
    import dataclasses
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import add_free_bodies, quat_mul, self_collision_variant
    from uhc_amd.model.shapes import box_triangles
    from uhc_amd.sim import make_ctrl
    if os.environ.get("UHC_FORCE_GENERAL") == "1":
        pytest.skip("the batch has no fast tier to start from")
    K = 7
    m = self_collision_variant(model)
    yaw = [0.06 * (-1) ** k for k in range(K)]
    poses = np.array([[1.0 + 0.305 * k, 1.0 + 0.01 * k, 0.1495, np.cos(y / 2), 0, 0, np.sin(y / 2)] for k, y in enumerate(yaw)], dtype=np.float64)
    m = dataclasses.replace(add_free_bodies(m, [box_triangles(0.15, 0.15, 0.15)] * K, poses, density=5.0 / 0.027), solver=1)
    ctrl = make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
    n = 12
    rng = np.random.default_rng(5)
    q = np.tile(m.qpos0, (n, 1))
    for e in range(n):
        qh = standing["qpos"].copy()
        qh[7:] += rng.normal(scale=0.01, size=69)
        if e in (3, 9):  # face down beside the raft: 300+ rows
            a = np.pi / 2 + 0.05 * e
            qh[3:7] = quat_mul(np.array([np.cos(a / 2), 0, np.sin(a / 2), 0]), qh[3:7])
            qh[0], qh[1], qh[2] = -0.8, -0.8, 0.14
        elif e == 5:
            qh[2] += 30.0  # airborne: fast tier
        q[e, :76] = qh
    v = np.zeros((n, m.nv))
    b = S.SimBatch(m, ctrl, n)
    b.set_kernel_path(2)
    tb = torch.zeros(n, 69, dtype=torch.float64, device="cuda")
    act = np.zeros((n, ctrl.action_dim))
    os_ = [OracleSim(m, ctrl) for _ in range(n)]
    primal, worst = 0, 0.0
    for t in range(6):
        b.set_state(torch.from_numpy(q), torch.from_numpy(v))
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        gq = b.field(S.F_QPOS).cpu().numpy()
        redo = b.field(S.F_REDO).cpu().numpy()
        tiers = b.field(S.F_TIER).cpu().numpy()
        for e in range(n):
            os_[e].set_state(q[e], v[e])
            os_[e].do_simulation(act[e], np.zeros(69))
            worst = max(worst, np.abs(gq[e] - os_[e].get("qpos")).max())
            primal += int((redo[e] & (1 << 30)) != 0)
        assert not (redo & 0x80).any() and not (redo & 2).any(), [hex(int(x)) for x in redo]
    print(f"sticky queues + tier 4: env-steps through tier 4 {primal} / {6 * n}; next-step tiers {tiers.tolist()}; worst |dqpos| {worst:.2e}")
    assert primal >= 6 and worst < 1e-9, (primal, worst)
    assert int(b.field(S.F_EFC_OVERFLOW).sum().item()) == 0 and int(b.field(S.F_FAIL).sum().item()) == 0


def test_tier_4_consumers_take_what_the_large_tiers_consumers_hand_on(model, standing, monkeypatch, capfd):
    """Kernel path 2 with ALL the queues running: ten envs standing with the boxes far above them (kept in the general tier by UHC_TIER_MARKS, so that its
    queue has envs of its own and the consumers of every tier are launched), two face down beside the raft on the floor (large tier by stickiness; its
    consumer finds them too big in the first substep and appends them to tier 4's queue, whose consumers -- uhc_k_huge_q.hip -- take them up while the
    other launches still run).  Every env re-posed each step; against the oracle, step by step; the library's own log (UHC_DEBUG bit 6) says whether
    the tier-4 consumers were launched."""
    import dataclasses
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import add_free_bodies, quat_mul, self_collision_variant
    from uhc_amd.model.shapes import box_triangles
    from uhc_amd.sim import make_ctrl
    if os.environ.get("UHC_FORCE_GENERAL") == "1" or os.environ.get("UHC_TIERS") in ("2", "3"):
        pytest.skip("needs the fast tier to start from and tier 4 to end in")
    monkeypatch.setenv("UHC_DEBUG", "64")
    monkeypatch.setenv("UHC_TIER_MARKS", "8,2,1,4,1,0,8,7")  # up to the general tier beyond 8 rows / 2 contacts / 1 body-body row; down again only below 4 / 1 / 0
    K = 7
    m = self_collision_variant(model)
    yaw = [0.06 * (-1) ** k for k in range(K)]
    poses = np.array([[1.0 + 0.305 * k, 1.0 + 0.01 * k, 0.1495, np.cos(y / 2), 0, 0, np.sin(y / 2)] for k, y in enumerate(yaw)], dtype=np.float64)
    m = dataclasses.replace(add_free_bodies(m, [box_triangles(0.15, 0.15, 0.15)] * K, poses, density=5.0 / 0.027), solver=1)
    ctrl = make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
    n, heavy = 12, (3, 9)
    rng = np.random.default_rng(7)
    q = np.tile(m.qpos0, (n, 1))
    for e in range(n):
        qh = standing["qpos"].copy()
        qh[7:] += rng.normal(scale=0.01, size=69)
        if e in heavy:  # face down beside the raft: 300+ rows
            a = np.pi / 2 + 0.05 * e
            qh[3:7] = quat_mul(np.array([np.cos(a / 2), 0, np.sin(a / 2), 0]), qh[3:7])
            qh[0], qh[1], qh[2] = -0.8, -0.8, 0.14
        else:  # standing, the boxes 5-11 m above the floor: two feet on the ground, nothing else
            for k in range(K):
                q[e, 76 + 7 * k + 2] = 5.0 + k
        q[e, :76] = qh
    v = np.zeros((n, m.nv))
    b = S.SimBatch(m, ctrl, n)
    b.set_kernel_path(2)
    tb = torch.zeros(n, 69, dtype=torch.float64, device="cuda")
    act = np.zeros((n, ctrl.action_dim))
    os_ = [OracleSim(m, ctrl) for _ in range(n)]
    primal, worst, tiers_seen = 0, 0.0, []
    for t in range(8):
        b.set_state(torch.from_numpy(q), torch.from_numpy(v))
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        gq = b.field(S.F_QPOS).cpu().numpy()
        redo = b.field(S.F_REDO).cpu().numpy()
        tiers_seen.append(b.field(S.F_TIER).cpu().numpy().tolist())
        for e in range(n):
            os_[e].set_state(q[e], v[e])
            os_[e].do_simulation(act[e], np.zeros(69))
            worst = max(worst, np.abs(gq[e] - os_[e].get("qpos")).max())
            primal += int((redo[e] & (1 << 30)) != 0)
            assert bool(redo[e] & (1 << 30)) == (e in heavy), (t, e, hex(int(redo[e])))
        assert not (redo & 0x80).any() and not (redo & 2).any(), [hex(int(x)) for x in redo]
    err = capfd.readouterr().err
    launched = [ln for ln in err.splitlines() if "tier-4 consumers" in ln]
    print(f"all queues + tier 4's consumers: {len(launched)} of 8 steps had them ({launched[:1]}); env-steps through tier 4 {primal}; next-step tiers {tiers_seen[-1]}; worst |dqpos| {worst:.2e}")
    assert len(launched) >= 4, (err[-2000:], tiers_seen)
    assert primal == 8 * len(heavy) and worst < 1e-9, (primal, worst)
    assert int(b.field(S.F_EFC_OVERFLOW).sum().item()) == 0 and int(b.field(S.F_FAIL).sum().item()) == 0
    b.close()


def test_a_dof_chain_of_32_entries_goes_through_tier_4(monkeypatch, capfd):
    """uhc_batch_create accepts dof chains of maxdepth + 1 == 32 entries, the SMPL models stop at 30 -- and tier 4's pair tables (528 pairs; the four
    ownership classes of the four-wave pass) were sized by them.  A synthetic 27-segment chain on the floor: depth 32, 432 rows (beyond the large tier), against the oracle's
    own primal solver -- through the chained launches (tier 4 on one wave: the 528-entry pair table) and, in a batch with two half-lifted animals that live in the general tier so
    that every tier's consumers are launched, through tier 4's four-wave consumers (the class tables at len 32)."""
    import dataclasses
    import torch
    from tests.helpers import caterpillar_model
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    if os.environ.get("UHC_FORCE_GENERAL") == "1" or os.environ.get("UHC_TIERS") in ("2", "3"):
        pytest.skip("needs the fast tier to start from and tier 4 to end in")
    m = dataclasses.replace(caterpillar_model(27), solver=1)
    assert m.nv == 32 and int(max(m.dof_depth) if hasattr(m, "dof_depth") else 31) + 1 == 32
    ctrl = passive_ctrl(m, n_substeps=15)
    rng = np.random.default_rng(3)

    def pose(lifted):
        q = m.qpos0.copy()
        q[2] -= 0.0004  # resting a little inside the floor's margin: every bottom vertex within it
        q[7:] = rng.normal(scale=1e-4, size=m.nq - 7)
        if lifted:  # the first six segments on the floor (96 rows: the general tier), the rest lifted off it
            q[7 + 5] = -1.1
        return q

    # ---- chained launches: tier 4 behind the large tier's workgroup, one wave
    n = 2
    q = np.stack([pose(False) for _ in range(n)])
    v = np.zeros((n, m.nv))  # (at rest: a velocity of a few cm/s lifts the animal out of the floor's margin within a step, and the lower tiers take it)
    b = S.SimBatch(m, ctrl, n)
    b.set_state(torch.from_numpy(q), torch.from_numpy(v))
    b.sync()
    os_ = [OracleSim(m, ctrl) for _ in range(n)]
    for e in range(n):
        os_[e].set_state(q[e], v[e])
        assert os_[e].geti("nefc") > 256 and int(b.field(S.F_NEFC)[e].item()) == os_[e].geti("nefc")
        np.testing.assert_allclose(b.field(S.F_QACC)[e].cpu().numpy(), os_[e].get("qacc"), atol=1e-7 * (1 + np.abs(os_[e].get("qacc")).max()))
    act = np.zeros((n, ctrl.action_dim))
    tb = torch.zeros(n, max(m.nu, 1), dtype=torch.float64, device="cuda")
    worst = 0.0
    for t in range(3):
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        redo = b.field(S.F_REDO).cpu().numpy()
        assert ((redo >> 30) & 1).all() and not (redo & 0x80).any() and not (redo & 2).any() and not ((redo >> 29) & 1).any(), [hex(int(x)) for x in redo]
        for e in range(n):
            os_[e].do_simulation(act[e], np.zeros(max(m.nu, 1)))
            worst = max(worst, np.abs(b.field(S.F_QPOS)[e].cpu().numpy() - os_[e].get("qpos")).max(), np.abs(b.field(S.F_QVEL)[e].cpu().numpy() - os_[e].get("qvel")).max())
    assert worst < 1e-8, worst
    b.close()
    # ---- sticky queues with every tier's consumers: the flat animals through tier 4's four-wave consumers
    monkeypatch.setenv("UHC_DEBUG", "64")
    n, flat = 6, (1, 4)
    q = np.stack([pose(e not in flat) for e in range(n)])
    v = np.zeros((n, m.nv))
    b = S.SimBatch(m, ctrl, n)
    b.set_kernel_path(2)
    act = np.zeros((n, ctrl.action_dim))
    tb = torch.zeros(n, max(m.nu, 1), dtype=torch.float64, device="cuda")
    os_ = [OracleSim(m, ctrl) for _ in range(n)]
    worst2, primal = 0.0, 0
    for t in range(8):  # every env re-posed each step, so that the scene stays what it is
        b.set_state(torch.from_numpy(q), torch.from_numpy(v))
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        redo = b.field(S.F_REDO).cpu().numpy()
        assert not (redo & 0x80).any() and not (redo & 2).any() and not ((redo >> 29) & 1).any(), [hex(int(x)) for x in redo]
        for e in range(n):
            os_[e].set_state(q[e], v[e])
            os_[e].do_simulation(act[e], np.zeros(max(m.nu, 1)))
            worst2 = max(worst2, np.abs(b.field(S.F_QPOS)[e].cpu().numpy() - os_[e].get("qpos")).max())
            primal += int((redo[e] >> 30) & 1)
            assert bool((redo[e] >> 30) & 1) == (e in flat), (t, e, hex(int(redo[e])), int(b.field(S.F_NEFC)[e].item()))
    err = capfd.readouterr().err
    launched = [ln for ln in err.splitlines() if "tier-4 consumers" in ln]
    print(f"32-entry dof chains: chained launches worst {worst:.2e}; sticky queues: {len(launched)} of 8 steps with tier-4 consumers, {primal} env-steps through tier 4, worst |dqpos| {worst2:.2e}")
    assert len(launched) >= 3, err[-1500:]
    assert primal == 8 * len(flat) and worst2 < 1e-9, (primal, worst2)
    b.close()
