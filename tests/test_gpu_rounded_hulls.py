"""Spheres and capsules on the device (round 6): the HIP step kernels through the C-ABI against oracle/physics_oracle.c.  The rounded hulls are what the reference's
masterfoot bodies are made of (capsule geoms, uhc/smpllib/smpl_robot.py:1386-1392): plane-capsule / plane-sphere contacts (k_collision's plane pass: core vertex
lowered by the radius, frame along the capsule) and rounded hull against mesh through MPR (support = core vertex + (margin / 2 + radius) dir).  Bar: contact counts
equal, accelerations 1e-6, trajectories 1e-8 (the solvers stop at tolerances)."""
import dataclasses
import os

import numpy as np
import pytest

from tests.helpers import box_triangles, passive_ctrl

pytestmark = pytest.mark.gpu

# a "foot": a free box (mesh) with three capsule toes on hinges, beside it a free sphere and a free capsule; the toes have contype 0 / conaffinity 1 like the
# reference's masterfoot geoms (they meet the floor and the other bodies' hulls, not each other)
FOOT_XML = """
<mujoco>
  <compiler angle="radian" coordinate="local" inertiafromgeom="true"/>
  <option timestep="0.002"/>
  <default><geom condim="3" margin="0.001"/></default>
  <asset><mesh name="foot" file="unused.stl"/><mesh name="brick" file="unused.stl"/></asset>
  <worldbody>
    <geom name="floor" type="plane" size="20 20 0.1"/>
    <body name="foot" pos="0 0 0.3">
      <joint name="root" type="free"/>
      <geom type="mesh" mesh="foot"/>
      <body name="toe0" pos="0.1 -0.05 -0.01">
        <joint name="t0y" type="hinge" axis="0 1 0" pos="0 0 0" limited="true" range="-0.5 0.5"/>
        <joint name="t0z" type="hinge" axis="0 0 1" pos="0 0 0" limited="true" range="-0.5 0.5"/>
        <geom type="capsule" size="0.02" fromto="0.02 0 0 0.09 0 0" contype="0" conaffinity="1"/>
      </body>
      <body name="toe1" pos="0.1 0 -0.01">
        <joint name="t1y" type="hinge" axis="0 1 0" pos="0 0 0" limited="true" range="-0.5 0.5"/>
        <geom type="capsule" size="0.02" fromto="0.02 0 0 0.1 0 0" contype="0" conaffinity="1"/>
      </body>
      <body name="toe2" pos="0.1 0.05 -0.01">
        <joint name="t2y" type="hinge" axis="0 1 0" pos="0 0 0" limited="true" range="-0.5 0.5"/>
        <geom type="capsule" size="0.02" fromto="0.02 0 0 0.08 0 0" contype="0" conaffinity="1"/>
      </body>
    </body>
    <body name="ball" pos="0.5 0 0.3">
      <joint type="free"/>
      <geom type="sphere" size="0.06"/>
    </body>
    <body name="rod" pos="-0.5 0 0.3">
      <joint type="free"/>
      <geom type="capsule" size="0.03" fromto="-0.1 0 0 0.1 0 0"/>
    </body>
    <body name="brick" pos="0 0.6 0.3">
      <joint type="free"/>
      <geom type="mesh" mesh="brick"/>
    </body>
  </worldbody>
</mujoco>
"""


def foot_model():
    from uhc_amd.model.mjcf import compile_mjcf
    return dataclasses.replace(compile_mjcf(FOOT_XML, meshes={"foot": box_triangles(0.1, 0.08, 0.03), "brick": box_triangles(0.08, 0.05, 0.04)}), solver=1)


@pytest.fixture(params=["fast", "general"], autouse=True)
def kernel_path(request):
    old = os.environ.get("UHC_FORCE_GENERAL")
    os.environ["UHC_FORCE_GENERAL"] = "1" if request.param == "general" else "0"
    yield request.param
    if old is None:
        os.environ.pop("UHC_FORCE_GENERAL", None)
    else:
        os.environ["UHC_FORCE_GENERAL"] = old


def _poses(m, n, seed):
    """The four trees in random poses: on the floor or a little above it, and -- in every second env -- the ball, the rod and the brick ON or AGAINST the foot and
    each other, so that rounded hull meets mesh and rounded hull meets rounded hull through MPR."""
    from scipy.spatial.transform import Rotation as sR
    rng = np.random.default_rng(seed)
    q = np.tile(m.qpos0, (n, 1))
    quat = lambda rv: np.roll(sR.from_rotvec(rv).as_quat(), 1)
    for e in range(n):
        # foot: level or tilted, toes bent
        q[e, 2] = 0.03 + rng.uniform(-0.0008, 0.01)
        q[e, 3:7] = quat(rng.normal(size=3) * (0.02 if e % 3 else 0.3))
        q[e, 7:11] = rng.uniform(-0.4, 0.4, size=4)
        ball, rod, brick = 11, 18, 25
        if e % 2 == 0:  # apart, each on the floor
            q[e, ball:ball + 3] = [0.5, 0, 0.06 + rng.uniform(-0.0008, 0.005)]
            q[e, rod:rod + 3] = [-0.5, 0, 0.03 + rng.uniform(-0.0008, 0.005)]
            q[e, rod + 3:rod + 7] = quat(np.array([0, rng.normal() * 0.05, rng.normal()]))
            q[e, brick:brick + 3] = [0, 0.6, 0.04 + rng.uniform(-0.0008, 0.005)]
            q[e, brick + 3:brick + 7] = quat(np.array([0, 0, rng.normal()]))
        else:  # piled up around the foot
            q[e, ball:ball + 3] = [rng.uniform(-0.05, 0.05), rng.uniform(-0.04, 0.04), q[e, 2] + 0.03 + 0.06 + rng.uniform(-0.004, 0.001)]
            q[e, rod:rod + 3] = [0.15 + rng.uniform(-0.02, 0.02), rng.uniform(-0.03, 0.03), q[e, 2] + 0.01 + 0.05 + rng.uniform(-0.004, 0.002)]
            q[e, rod + 3:rod + 7] = quat(np.array([0, 0, np.pi / 2 + rng.normal() * 0.2]))
            q[e, brick:brick + 3] = [q[e, ball] + rng.uniform(-0.03, 0.03), q[e, ball + 1] + 0.06 + 0.05 + rng.uniform(-0.004, 0.001), q[e, ball + 2] + rng.uniform(-0.02, 0.02)]
            q[e, brick + 3:brick + 7] = quat(rng.normal(size=3) * 0.2)
    return q


def test_rounded_hull_contacts_match_oracle(kernel_path):
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import GEOM_CAPSULE, GEOM_SPHERE
    m = foot_model()
    assert m.nv == 28 and (m.geom_type == GEOM_CAPSULE).sum() == 4 and (m.geom_type == GEOM_SPHERE).sum() == 1
    n = 64
    q = _poses(m, n, 5)
    v = np.random.default_rng(6).normal(scale=0.2, size=(n, m.nv))
    b = S.SimBatch(m, passive_ctrl(m), n)
    b.set_state(torch.from_numpy(q), torch.from_numpy(v))
    b.sync()
    ncon, nefc, qacc = (b.field(f).cpu().numpy() for f in (S.F_NCON, S.F_NEFC, S.F_QACC))
    plane_round = round_mesh = round_round = 0
    for e in range(n):
        o = OracleSim(m)
        o.desc.solver = 0 if (int(b.field(S.F_REDO)[e].item()) & 2) else 1
        o.set_state(q[e], v[e])
        assert ncon[e] == o.geti("ncon") and nefc[e] == o.geti("nefc"), (e, ncon[e], o.geti("ncon"), nefc[e], o.geti("nefc"))
        np.testing.assert_allclose(qacc[e], o.get("qacc"), atol=1e-6 * (1 + np.abs(o.get("qacc")).max()), rtol=1e-6)
        g1, g2 = o.get("con_geom1").astype(int), o.get("con_geom2").astype(int)
        for a, c in zip(g1, g2):
            ra, rc = m.geom_type[a] in (GEOM_SPHERE, GEOM_CAPSULE), m.geom_type[c] in (GEOM_SPHERE, GEOM_CAPSULE)
            plane_round += int(m.geom_type[a] == 0 and rc)
            round_mesh += int(a != 0 and ra != rc)
            round_round += int(ra and rc)
    print(f"rounded hulls [{kernel_path}]: {plane_round} plane-round, {round_mesh} round-mesh, {round_round} round-round contacts over {n} envs")
    assert plane_round > 60 and round_mesh > 15 and round_round > 3
    b.close()


def test_rounded_hull_trajectories_match_oracle(kernel_path):
    """Forty control steps (5 substeps each) from the piled-up and the apart poses: balls roll, rods tip over, toes bend under the foot.  The oracle is re-started
    from the device's state before every control step (a rolling ball on a brick is a chaotic scene: free-running copies part within ten steps): every step within
    1e-8 -- except steps at which the contact model itself is discontinuous: a capsule lying FLAT on a mesh face gives MPR a support query whose two candidates (the
    ends of the segment) tie to the last bit, rounding picks the end, and the contact lands somewhere else along the capsule (libccd's MPR does the same; MuJoCo too).
    Such a step is recognised by the oracle's own sensitivity (its answer moves by > 1e-8 when started 1e-14 away), and the device must then have landed on one of
    the oracle's own branches or within twice their spread (the rule of tests/test_gpu_parity_200.py)."""
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    m = foot_model()
    n = 16
    q = _poses(m, n, 11)
    v = np.random.default_rng(12).normal(scale=0.3, size=(n, m.nv))
    ctrl = passive_ctrl(m, n_substeps=5)
    b = S.SimBatch(m, ctrl, n)
    b.set_state(torch.from_numpy(q), torch.from_numpy(v))
    b.sync()
    os_ = [OracleSim(m, ctrl) for _ in range(n)]
    act = torch.zeros(n, ctrl.action_dim, dtype=torch.float64, device="cuda")
    tb = torch.zeros(n, max(m.nu, 1), dtype=torch.float64, device="cuda")
    za, zt = np.zeros(ctrl.action_dim), np.zeros(max(m.nu, 1))
    worst, most, ties = 0.0, 0, []
    for t in range(40):
        pq, pv = b.field(S.F_QPOS).cpu().numpy().copy(), b.field(S.F_QVEL).cpu().numpy().copy()
        b.simulate(act, tb)
        b.sync()
        gq, gv = b.field(S.F_QPOS).cpu().numpy(), b.field(S.F_QVEL).cpu().numpy()
        redo = b.field(S.F_REDO).cpu().numpy()
        for e in range(n):
            o = os_[e]
            o.desc.solver = 0 if (int(redo[e]) & 2) else 1
            o.set_state(pq[e], pv[e])
            o.do_simulation(za, zt)
            err = max(np.abs(gq[e] - o.get("qpos")).max(), np.abs(gv[e] - o.get("qvel")).max())
            most = max(most, o.geti("nefc"))
            if err > 1e-8:
                sens, near = 0.0, err
                for k_ in (1, 2, 3):
                    for s_ in (1.0, -1.0):
                        p2 = pq[e].copy()
                        lin = np.r_[0:3, 7:11, 11:14, 18:21, 25:28]  # every tree's position and the toes' angles (the quaternions stay unit)
                        p2[lin] += s_ * 1e-14 * np.cos(k_ * np.arange(lin.size))
                        twin = OracleSim(m, ctrl)
                        twin.desc.solver = o.desc.solver
                        twin.set_state(p2, pv[e])
                        twin.do_simulation(za, zt)
                        sens = max(sens, np.abs(twin.get("qpos") - o.get("qpos")).max(), np.abs(twin.get("qvel") - o.get("qvel")).max())
                        near = min(near, max(np.abs(gq[e] - twin.get("qpos")).max(), np.abs(gv[e] - twin.get("qvel")).max()))
                ties.append((t, e, err, sens, near))
                assert sens > 1e-8 and (near < 1e-7 or err <= 2.0 * sens), f"step {t} env {e}: device {err:.2e} from the oracle, whose own spread is {sens:.2e} (nearest branch {near:.2e})"
                err = 0.0
            worst = max(worst, err)
    print(f"rounded hulls [{kernel_path}]: 40 control steps x {n} envs, worst |d(qpos, qvel)| of a control step {worst:.2e}, most rows {most}; {len(ties)} steps at a tie of the contact "
          f"model (oracle's own spread {min((x[3] for x in ties), default=0):.1e} .. {max((x[3] for x in ties), default=0):.1e})")
    assert worst < 1e-8, worst
    assert len(ties) <= 0.15 * 40 * n, len(ties)
    assert int(b.field(S.F_FAIL).sum().item()) == 0
    b.close()
