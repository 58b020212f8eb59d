"""GPU parity for the ball-joint humanoid (robot.ball: True, config/copycat_ball/copycat_ball_1.yml:99: one ball joint + three
gear-vector motors per bone, action_type torque) and for free objects around it (BASELINE configs[4] stand-in):
nq = 99 (+7 per object), two-tree contacts, self-collision on.  HIP path through the C-ABI vs the CPU oracle."""
import dataclasses
import os

import numpy as np
import pytest

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


def _ball_setup(model, objects=0, self_collision=False):
    from tests.helpers import box_triangles
    from uhc_amd.model.mjcf import add_free_bodies, ball_variant, self_collision_variant
    from uhc_amd.sim import make_ctrl
    ball = ball_variant(model)
    if self_collision:
        ball = self_collision_variant(ball)
    if objects:
        rng = np.random.default_rng(11)
        ang = rng.uniform(0, 2 * np.pi, size=objects)
        poses = np.stack([np.r_[-0.15 + 0.7 * np.cos(a), -0.05 + 0.7 * np.sin(a), 0.16 + 0.4 * k, 1, 0, 0, 0] for k, a in enumerate(ang)])
        ball = add_free_bodies(ball, [box_triangles(0.15, 0.15, 0.15)] * objects, poses, density=5.0 / 0.027)
    ball = dataclasses.replace(ball, solver=1)
    hinge_ctrl = make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)  # copycat_ball_1.yml: torque, tq_mul 4, no RFC
    return ball, hinge_ctrl


def _states(model, ball, standing, n, seed, lift):
    from uhc_amd.model.mjcf import hinge_to_ball_qpos, ball_variant
    rng = np.random.default_rng(seed)
    base = ball_variant(model)
    q = np.tile(ball.qpos0, (n, 1))
    for e in range(n):
        qh = standing["qpos"].copy()
        qh[7:] += rng.normal(scale=0.1, size=69)
        qh[2] += lift
        q[e, :99] = hinge_to_ball_qpos(model, base, qh)
    v = np.zeros((n, ball.nv))
    v[:, :75] = rng.normal(scale=0.3, size=(n, 75))
    return q, v


@pytest.mark.parametrize("lift", [3.0, 0.0])
def test_ball_humanoid_trajectory_matches_oracle(model, standing, kernel_path, lift):
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    ball, ctrl = _ball_setup(model)
    n = 4
    q, v = _states(model, ball, standing, n, 41, lift)
    b = S.SimBatch(ball, ctrl, n)
    b.set_state(torch.from_numpy(q), torch.from_numpy(v))
    b.sync()
    os_ = [OracleSim(ball, ctrl) for _ in range(n)]
    for e in range(n):
        os_[e].desc.solver = 0 if (int(b.field(S.F_REDO)[e].item()) & 2) else 1
        os_[e].set_state(q[e], v[e])
        np.testing.assert_allclose(b.field(S.F_QM)[e].cpu().numpy(), os_[e].get("qM"), atol=1e-10)
        np.testing.assert_allclose(b.field(S.F_QACC)[e].cpu().numpy(), os_[e].get("qacc"), atol=1e-5, rtol=1e-6)
    for o in os_:
        o.desc.solver = 1  # (the forward pass of set_state may have been compared under solver 0, see above)
    rng = np.random.default_rng(42)
    tb = torch.zeros(n, 69, dtype=torch.float64, device="cuda")
    worst = 0.0
    for t in range(15):
        act = rng.normal(scale=0.003, size=(n, ctrl.action_dim))  # x a_scale x 100: torques of a few N m (the hands weigh 0.4 kg)
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        gq = b.field(S.F_QPOS).cpu().numpy()
        redo = b.field(S.F_REDO).cpu().numpy()
        for e in range(n):
            os_[e].do_simulation(act[e], np.zeros(69), redo=redo[e])  # UHC_F_REDO bits 8+: the substeps the general kernel solved by sweeps
            worst = max(worst, np.abs(gq[e] - os_[e].get("qpos")).max())
    assert worst < 1e-6, worst
    assert np.abs(np.linalg.norm(gq[:, 7:11], axis=1) - 1).max() < 1e-12  # ball quaternions stay normalised


def test_ball_humanoid_with_objects_and_self_collision(model, standing, kernel_path):
    """configs[4] stand-in: ball joints, self-collision, 3 free boxes (5 kg, 0.3 m) next to the humanoid: nq 120, nv 93, three trees."""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    ball, ctrl = _ball_setup(model, objects=3, self_collision=True)
    assert (ball.nq, ball.nv, ball.nbody) == (99 + 21, 75 + 18, 28)
    n = 3
    q, v = _states(model, ball, standing, n, 43, 0.0)
    b = S.SimBatch(ball, ctrl, n)
    b.set_state(torch.from_numpy(q), torch.from_numpy(v))
    b.sync()
    os_ = [OracleSim(ball, ctrl) for _ in range(n)]
    redo = b.field(S.F_REDO).cpu().numpy()
    for e in range(n):
        os_[e].desc.solver = 0 if (redo[e] & 2) else 1
        os_[e].set_state(q[e], v[e])
        assert int(b.field(S.F_NCON)[e].item()) == os_[e].geti("ncon") and int(b.field(S.F_NEFC)[e].item()) == os_[e].geti("nefc")
    for o in os_:
        o.desc.solver = 1  # (the forward pass of set_state may have been compared under solver 0, see above)
    rng = np.random.default_rng(44)
    tb = torch.zeros(n, 69, dtype=torch.float64, device="cuda")
    worst = 0.0
    for t in range(12):
        act = rng.normal(scale=0.003, size=(n, ctrl.action_dim))  # x a_scale x 100: torques of a few N m (the hands weigh 0.4 kg)
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        gq = b.field(S.F_QPOS).cpu().numpy()
        redo = b.field(S.F_REDO).cpu().numpy()
        for e in range(n):
            os_[e].do_simulation(act[e], np.zeros(69), redo=redo[e])  # UHC_F_REDO bits 8+: the substeps the general kernel solved by sweeps
            worst = max(worst, np.abs(gq[e] - os_[e].get("qpos")).max())
    assert worst < 1e-5, worst
    assert int(b.field(S.F_FAIL).sum().item()) == 0


@pytest.mark.parametrize("self_collision", [False, True], ids=["floor_only", "self_collision"])
def test_ball_joint_limits_match_oracle(model, standing, kernel_path, self_collision):
    """[MJ-ext] mj_instantiateLimit's ball branch on the device (k_enumerate_rows / k_rows / k_rows_fast: one row per ball joint whose rotation
    angle comes within the margin of max(range), Jacobian -axis on its three dofs) against the oracle's (pinned by the cone KAT in
    tests/test_oracle_physics.py).  The generated ball humanoid carries no ranges (uhc/khrylib/mocap/skeleton_mesh_v2.py:256-267 writes
    none); a model that does bounds how far a torque policy can fold it.  The floor-only variant runs the fast tier's DENSE
    instantiation too (KernelArgs::ball_limits): the limit rows live there."""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import JNT_BALL
    ball, ctrl = _ball_setup(model, self_collision=self_collision)
    ball = ball.copy()
    isb = np.asarray(ball.jnt_type) == JNT_BALL
    ball.jnt_limited = isb.astype(np.int32)
    # every joint's cone = its angle in the standing pose + 0.05 rad; the states add 0.1 rad of noise per hinge axis: some joints start beyond it
    from uhc_amd.model.mjcf import hinge_to_ball_qpos
    q_stand = hinge_to_ball_qpos(model, ball, standing["qpos"])[7:99].reshape(23, 4)
    ang0 = 2 * np.arctan2(np.linalg.norm(q_stand[:, 1:], axis=1), np.abs(q_stand[:, 0]))
    rng_ = np.zeros((ball.njnt, 2))
    rng_[isb, 1] = ang0 + 0.05
    ball.jnt_range = rng_
    n = 4
    q, v = _states(model, ball, standing, n, 47, 0.3)
    b = S.SimBatch(ball, ctrl, n)
    b.set_state(torch.from_numpy(q), torch.from_numpy(v))
    b.sync()
    os_ = [OracleSim(ball, ctrl) for _ in range(n)]
    redo = b.field(S.F_REDO).cpu().numpy()
    for e in range(n):
        os_[e].desc.solver = 0 if (redo[e] & 2) else 1
        os_[e].set_state(q[e], v[e])
        assert int(b.field(S.F_NEFC)[e].item()) == os_[e].geti("nefc")
        np.testing.assert_allclose(b.field(S.F_QACC)[e].cpu().numpy(), os_[e].get("qacc"), atol=1e-5, rtol=1e-6)
    assert min(o.geti("nefc") for o in os_) >= 3  # airborne (lift 0.3): every row is a ball-joint limit
    for o in os_:
        o.desc.solver = 1
    rng = np.random.default_rng(48)
    tb = torch.zeros(n, 69, dtype=torch.float64, device="cuda")
    worst = 0.0
    for t in range(12):
        act = rng.normal(scale=0.003, size=(n, ctrl.action_dim))
        b.simulate(torch.from_numpy(act).cuda(), tb)
        b.sync()
        gq = b.field(S.F_QPOS).cpu().numpy()
        redo = b.field(S.F_REDO).cpu().numpy()
        for e in range(n):
            os_[e].do_simulation(act[e], np.zeros(69), redo=redo[e])
            worst = max(worst, np.abs(gq[e] - os_[e].get("qpos")).max())
    assert worst < 1e-6, worst
    # the limits hold: no joint ends far beyond its cone (soft limits: a few hundredths of a radian in flight, up to ~0.2 under the landing's contact forces)
    ang = 2 * np.arctan2(np.linalg.norm(gq[:, 7:99].reshape(n, 23, 4)[..., 1:], axis=-1), np.abs(gq[:, 7:99].reshape(n, 23, 4)[..., 0]))
    assert (ang - (ang0 + 0.05)[None]).max() < 0.35
    assert int(b.field(S.F_FAIL).sum().item()) == 0
    b.close()
