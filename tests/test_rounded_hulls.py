"""Spheres and capsules (round 6): model compiler + oracle, on the CPU.  The reference's masterfoot variant hangs twelve CAPSULE bodies under each ankle
(uhc/smpllib/smpl_robot.py:1336-1414); MuJoCo collides them with the floor through mjc_PlaneCapsule (two sphere tests at the segment's ends) and with the body
hulls through mjc_Convex with a capsule support function.  Here both are ROUNDED HULLS (include/uhc_amd.h): core vertices + a radius.  MuJoCo is not in this image,
so these are known-answer tests of the restatement (analytic inertia, closed-form resting depth, geometry of the contacts), not a pin."""
import numpy as np
import pytest

from uhc_amd.model.mjcf import GEOM_CAPSULE, GEOM_SPHERE, compile_mjcf, export_mjcf, geom_radius, scale_model

CAPSULE_XML = """
<mujoco>
  <compiler angle="radian" coordinate="local" inertiafromgeom="true"/>
  <option timestep="0.002"/>
  <default><geom condim="3" margin="0.001"/></default>
  <worldbody>
    <geom name="floor" type="plane" size="20 20 0.1"/>
    <body name="cap" pos="0 0 0.5">
      <joint type="free"/>
      <geom name="c" type="capsule" size="{r}" fromto="{ft}"/>
    </body>
  </worldbody>
</mujoco>
"""


def capsule_model(r=0.035, ft="-0.05 0 0 0.05 0 0"):
    m = compile_mjcf(CAPSULE_XML.format(r=r, ft=ft))
    m.solver = 1
    return m


def test_capsule_mass_properties_are_the_solid_of_revolution():
    """[MJ-ext] mjCGeom::SetInertia for a capsule = cylinder + two half spheres displaced along the axis; against a Monte Carlo integral of the solid itself."""
    r, hl = 0.035, 0.05
    m = capsule_model(r)
    assert m.geom_type[1] == GEOM_CAPSULE and m.geom_vertnum[1] == 2 and geom_radius(m)[1] == r
    np.testing.assert_allclose(m.mesh_vert[m.geom_vertadr[1]:m.geom_vertadr[1] + 2], [[0.05, 0, 0], [-0.05, 0, 0]], atol=1e-15)  # pos + segment first
    vol = np.pi * r * r * 2 * hl + 4 / 3 * np.pi * r ** 3
    assert m.body_mass[1] == pytest.approx(1000 * vol, rel=1e-12)
    rng = np.random.default_rng(0)
    P = rng.uniform([-hl - r, -r, -r], [hl + r, r, r], size=(2_000_000, 3))
    x = np.clip(P[:, 0], -hl, hl)
    Q = P[(P[:, 0] - x) ** 2 + P[:, 1] ** 2 + P[:, 2] ** 2 <= r * r]
    I = m.body_mass[1] * np.array([(Q[:, 1] ** 2 + Q[:, 2] ** 2).mean(), (Q[:, 0] ** 2 + Q[:, 2] ** 2).mean(), (Q[:, 0] ** 2 + Q[:, 1] ** 2).mean()])
    np.testing.assert_allclose(np.sort(m.body_inertia[1]), np.sort(I), rtol=3e-3)
    assert m.geom_rbound[1] == pytest.approx(hl + r)
    # the exported file compiles back to the same model, and a scaled model scales the radius with the core
    m2 = compile_mjcf(export_mjcf(m, density=1000.0))
    np.testing.assert_allclose(m2.body_inertia, m.body_inertia, rtol=1e-12)
    np.testing.assert_allclose(m2.mesh_vert, m.mesh_vert, atol=1e-15)
    s = scale_model(m, 1.2)
    assert geom_radius(s)[1] == pytest.approx(1.2 * r) and s.body_mass[1] == pytest.approx(1.2 ** 3 * m.body_mass[1])


def test_capsule_on_the_floor_two_contacts_at_the_segment_ends_frame_along_the_axis():
    """mjc_PlaneCapsule: one sphere test per segment end -- dist = height of the end - radius, pos = end - n (radius + dist / 2), the contact frame's second axis
    along the capsule; an end further than the margin above the floor has no contact."""
    from oracle.physics import OracleSim
    r = 0.035
    m = capsule_model(r)
    s = OracleSim(m)
    yaw = 0.3
    q = np.array([0.2, -0.1, r - 0.0004, np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])
    s.set_state(q, np.zeros(6))
    assert s.geti("ncon") == 2 and s.geti("nefc") == 8
    ends = np.array([q[:3] + 0.05 * np.array([np.cos(yaw), np.sin(yaw), 0]), q[:3] - 0.05 * np.array([np.cos(yaw), np.sin(yaw), 0])])
    np.testing.assert_allclose(s.get("con_dist")[:2], [-0.0004, -0.0004], atol=1e-15)
    pos = s.get("con_pos").reshape(-1, 3)[:2]
    pos = pos[np.argsort(-pos[:, 0])]  # (the lower end comes first, and lying flat that is decided by rounding)
    np.testing.assert_allclose(pos, ends - np.array([0, 0, 1]) * (r + 0.5 * -0.0004), atol=1e-15)
    fr = s.get("con_frame").reshape(-1, 3, 3)[:2]
    for f in fr:
        np.testing.assert_allclose(f[0], [0, 0, 1], atol=1e-15)
        np.testing.assert_allclose(f[1], [np.cos(yaw), np.sin(yaw), 0], atol=1e-12)
        np.testing.assert_allclose(f[2], np.cross(f[0], f[1]), atol=1e-12)
    # tilted about y: the +x end rises (a rotation about y takes x towards -z for a positive angle: it is the -x end that rises) -- one contact only
    pitch = 0.1
    q = np.array([0, 0, r + 0.05 * np.sin(pitch) - 0.0004, np.cos(pitch / 2), 0, np.sin(pitch / 2), 0])
    s.set_state(q, np.zeros(6))
    assert s.geti("ncon") == 1
    assert s.get("con_dist")[0] == pytest.approx(-0.0004, abs=1e-12)
    assert s.get("con_pos")[0] == pytest.approx(0.05 * np.cos(pitch), abs=1e-12)  # under the +x end, which went down


def test_sphere_and_capsule_rest_at_the_solref_solimp_equilibrium():
    """The closed form of tests/test_oracle_physics.py::test_resting_penetration...: at rest every active pyramid edge carries k imp(r) (margin - r) / R_py; the
    edges of all contacts add up to the weight.  One contact (sphere), two (capsule lying flat)."""
    from oracle.physics import OracleSim
    mu, margin, tc, dr = 1.0, 0.001, 0.02, 1.0
    dmin, dmax, width, mid, power = 0.9, 0.95, 0.001, 0.5, 2.0
    k = 1.0 / (dmax * dmax * tc * tc * dr * dr)

    def imp(r):
        x = abs(r - margin) / width
        if x >= 1:
            return dmax
        y = (x / mid) ** power * mid if x <= mid else 1 - ((1 - x) / (1 - mid)) ** power * (1 - mid)
        return dmin + y * (dmax - dmin)

    sphere_xml = CAPSULE_XML.replace('type="capsule" size="{r}" fromto="{ft}"', 'type="sphere" size="0.05"')
    for m, ncon, h0 in ((compile_mjcf(sphere_xml), 1, 0.05), (capsule_model(), 2, 0.035)):
        m.solver = 1
        assert m.geom_type[1] in (GEOM_SPHERE, GEOM_CAPSULE)
        s = OracleSim(m)
        s.set_state(np.array([0, 0, h0 - 0.0001, 1, 0, 0, 0.0]), np.zeros(6))
        for _ in range(1500):
            s.step()
        s.forward()
        assert abs(s.get("qvel")).max() < 1e-6 and s.geti("ncon") == ncon
        f = s.get("efc_force")[:4 * ncon]
        assert f.sum() == pytest.approx(m.body_mass[1] * 9.81, rel=1e-6)
        tran = m.body_invweight0[1, 0] + m.body_invweight0[0, 0]

        def edge_force(r):
            i = imp(r)
            R0 = (1 - i) / i * (tran + mu * mu * tran)
            return k * i * (margin - r) / (2 * mu * mu * R0)

        lo, hi = -0.01, margin
        for _ in range(200):
            md = 0.5 * (lo + hi)
            if 4 * ncon * edge_force(md) > m.body_mass[1] * 9.81:
                lo = md
            else:
                hi = md
        dist = s.get("con_dist")[:ncon]
        assert np.ptp(dist) < 1e-9 and dist[0] == pytest.approx(0.5 * (lo + hi), abs=1e-9)
        assert s.get("qpos")[2] == pytest.approx(h0 + dist[0], abs=1e-9)


def test_mpr_between_rounded_hulls_is_the_distance_of_their_cores():
    """Two spheres, a sphere and a capsule, two crossed capsules: depth = r1 + r2 + margin - distance of the cores, normal along the shortest segment from hull 1 to
    hull 2, position midway between the two surfaces -- MPR's answer (to its 1e-6 tolerance) against the closed forms."""
    from oracle.physics import OracleSim
    xml = """
<mujoco>
  <compiler angle="radian" coordinate="local" inertiafromgeom="true"/>
  <default><geom condim="1" margin="0.002"/></default>
  <worldbody>
    <body name="a" pos="0 0 1"><joint type="free"/><geom type="{ta}" size="{sa}" {fa}/></body>
    <body name="b" pos="0 0 1"><joint type="free"/><geom type="{tb}" size="{sb}" {fb}/></body>
  </worldbody>
</mujoco>
"""
    cases = [
        # (geom a, geom b, pose of b relative to a, expected distance of the cores, expected normal)
        (("sphere", "0.05", ""), ("sphere", "0.03", ""), np.array([0.06, 0.02, 0.03, 1, 0, 0, 0.0]), None, None),
        (("sphere", "0.05", ""), ("capsule", "0.02", 'fromto="-0.1 0 0 0.1 0 0"'), np.array([0.03, 0.06, 0.0, 1, 0, 0, 0.0]), 0.06, np.array([0, 1.0, 0])),
        (("capsule", "0.03", 'fromto="-0.1 0 0 0.1 0 0"'), ("capsule", "0.02", 'fromto="0 -0.1 0 0 0.1 0"'), np.array([0.01, 0.02, 0.045, 1, 0, 0, 0.0]), 0.045, np.array([0, 0, 1.0])),
    ]
    for (ta, sa, fa), (tb, sb, fb), rel, d_exp, n_exp in cases:
        m = compile_mjcf(xml.format(ta=ta, sa=sa, fa=fa, tb=tb, sb=sb, fb=fb))
        s = OracleSim(m)
        q = np.r_[0, 0, 1, 1, 0, 0, 0.0, rel[:3] + [0, 0, 1], rel[3:]]
        s.set_state(q, np.zeros(12))
        assert s.geti("ncon") == 1
        r1, r2 = float(sa), float(sb)
        if d_exp is None:
            d_exp = np.linalg.norm(rel[:3]); n_exp = rel[:3] / d_exp
        dist = s.get("con_dist")[0]
        assert dist == pytest.approx(d_exp - r1 - r2, abs=2e-6)
        np.testing.assert_allclose(s.get("con_frame")[:3], n_exp, atol=2e-3)
        # midway between the two surfaces along the normal -- for two spheres; with a capsule in the pair MPR's position is the barycentre of portal points that lie on
        # different ends of the segment and on a curved surface (2.5 mm inside it here): libccd's own answer for such a pair, and not what MuJoCo computes for it
        # (sphere-sphere, sphere-capsule and capsule-capsule have analytic routines there; what goes through mjc_Convex is a rounded hull against a MESH, below)
        if ta == "sphere" and tb == "sphere":
            np.testing.assert_allclose(s.get("con_pos")[:3], np.array([0, 0, 1.0]) + n_exp * (r1 + 0.5 * (d_exp - r1 - r2)), atol=2e-4)


def test_capsule_against_a_mesh_face():
    """The pair the masterfoot toes need (capsule geoms with conaffinity 1 against the body hulls, uhc/smpllib/smpl_robot.py:1386-1392): a capsule lying across the top
    face of a box mesh.  Depth = margin + radius - height of the axis above the face, normal = the face normal (from the box to the capsule it is geom 1 -> geom 2), the
    position inside the overlap slab."""
    from oracle.physics import OracleSim
    from tests.helpers import box_triangles
    xml = """
<mujoco>
  <compiler angle="radian" coordinate="local" inertiafromgeom="true"/>
  <default><geom condim="1" margin="0.002"/></default>
  <asset><mesh name="box" file="unused.stl"/></asset>
  <worldbody>
    <body name="a" pos="0 0 1"><joint type="free"/><geom type="mesh" mesh="box"/></body>
    <body name="b" pos="0 0 1"><joint type="free"/><geom type="capsule" size="0.03" fromto="-0.05 0 0 0.05 0 0"/></body>
  </worldbody>
</mujoco>
"""
    m = compile_mjcf(xml, meshes={"box": box_triangles(0.1, 0.1, 0.1)})
    s = OracleSim(m)
    h = 0.1 + 0.03 - 0.004  # the axis 4 mm lower than touching
    yaw = 0.4
    q = np.r_[0, 0, 1, 1, 0, 0, 0.0, 0.01, -0.02, 1 + h, np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]
    s.set_state(q, np.zeros(12))
    assert s.geti("ncon") == 1
    assert s.get("con_dist")[0] == pytest.approx(-0.004, abs=2e-6)
    np.testing.assert_allclose(s.get("con_frame")[:3], [0, 0, 1], atol=1e-3)
    p = s.get("con_pos")[:3]
    assert 1.1 - 0.004 - 1e-3 <= p[2] <= 1.1 + 1e-3 and abs(p[0] - 0.01) < 0.06 and abs(p[1] + 0.02) < 0.03, p
    # lifted beyond radius + margin: apart
    q[9] = 1 + 0.1 + 0.03 + 0.0021
    s.set_state(q, np.zeros(12))
    assert s.geti("ncon") == 0


def _asset_body_provider(model):
    """The neutral asset's own hulls as 'SMPL vertices' (one-hot skin weights, joints = body origins): tests/test_model_compiler.py's synthetic body source."""
    from uhc_amd.model.mjcf import kinematics_np, quat_to_mat
    from uhc_amd.smpllib.smpl_mujoco import SMPL_BONE_ORDER_NAMES
    xpos, xquat, _, _ = kinematics_np(model, model.qpos0)
    verts, owner = [], []
    for g in range(model.ngeom):
        if model.geom_type[g] != 7:
            continue
        b = model.geom_bodyid[g]
        v = model.mesh_vert[model.geom_vertadr[g]:model.geom_vertadr[g] + model.geom_vertnum[g]] @ quat_to_mat(xquat[b]).T + xpos[b]
        verts.append(v)
        owner += [SMPL_BONE_ORDER_NAMES.index(model.body_names[b])] * len(v)
    verts = np.concatenate(verts)
    W = np.zeros((len(verts), 24))
    W[np.arange(len(verts)), owner] = 1
    joints = np.stack([xpos[model.body_names.index(n)] for n in SMPL_BONE_ORDER_NAMES])
    return lambda b, g: (verts, joints, W)


def test_masterfoot_model_is_generated_compiled_and_stepped_by_the_oracle(model, standing):
    """`robot: {masterfoot: true}` (config/masterfoot/*.yml of the reference; Robot.add_masterfoot, uhc/smpllib/smpl_robot.py:1336-1414): twelve capsule bodies with
    three hinges each under both ankles -- 49 bodies, 147 dofs, 141 motors.  Geometry of the toes against the reference's template arithmetic; the model compiles
    (capsule inertia, rounded hulls), the oracle stands it on the floor on its capsules; the HIP step kernels hold at most 128 dofs and the library says so."""
    from oracle.physics import OracleSim
    from uhc_amd.model.mjcf import GEOM_CAPSULE, kinematics_np
    from uhc_amd.smpllib.smpl_robot import Robot
    r = Robot({"mesh": True, "masterfoot": True, "master_range": 30}, body_provider=_asset_body_provider(model))
    r.load_from_skeleton(np.zeros(16), gender=[0])
    m = r.get_model()
    assert (m.nbody, m.nv, m.nq, m.nu) == (49, 147, 148, 141)
    caps = np.nonzero(m.geom_type == GEOM_CAPSULE)[0]
    assert len(caps) == 24 and (m.geom_contype[caps] == 0).all() and (m.geom_conaffinity[caps] == 1).all() and np.allclose(m.geom_size[caps, 0], 0.035)
    np.testing.assert_allclose(m.geom_size[caps, 1], 0.05, atol=1e-6)  # 0.1 long
    # the toes are children of the ankles, cloned at the ankle's origin with its three hinges, limited to +-30 degrees
    for side, first in (("L_Ankle", "L_Ankle_master0"), ("R_Ankle", "R_Ankle_master0")):
        a, b = m.body_names.index(side), m.body_names.index(first)
        assert m.body_parentid[b] == a and np.allclose(m.body_pos[b], 0) and m.body_jntnum[b] == 3
        j = m.body_jntadr[b]
        np.testing.assert_allclose(m.jnt_range[j:j + 3], np.tile(np.deg2rad([-30, 30]), (3, 1)), atol=1e-12)
    # the capsules' start points: the reference's template arithmetic, in the frame the XML is written in (SMPL's: y up)
    xpos, xquat, _, _ = kinematics_np(m, m.qpos0)
    ank, toe = xpos[m.body_names.index("L_Ankle")], xpos[m.body_names.index("L_Toe")]
    dm = np.linalg.norm(ank - toe) / 0.13432456960660616
    t = Robot.MASTERFOOT_TEMPLATE.copy()
    t[:, 2] -= 0.08 * dm; t[:, 0] += 0.05 * dm; t /= 3 / dm; t += ank
    g0 = [g for g in caps if m.geom_bodyid[g] == m.body_names.index("L_Ankle_master3")][0]
    start = m.mesh_vert[m.geom_vertadr[g0] + 1] + ank  # second core vertex = pos - half length along the axis = the `from` point (the toe's frame sits at the ankle)
    np.testing.assert_allclose(start[[0, 2]], t[3][[0, 2]], atol=2e-4)
    assert start[1] == pytest.approx(r.meshes["L_Ankle"].reshape(-1, 3)[:, 1].min(), abs=2e-4)
    # the oracle: upright on the floor (the standing clip's root pose), it comes to rest on capsules and hulls
    m.solver = 1
    s = OracleSim(m)
    q = m.qpos0.copy()
    q[:7] = standing["qpos"][:7]
    q[2] += 0.08  # (the capsules hang 3.5 cm below the soles' lowest vertices: start clear of the floor and let it land)
    s.set_state(q, np.zeros(m.nv))
    # (clear of the floor -- but not of itself: the capsules sit at the soles and overlap the toe hulls, their siblings under the ankle, whom MuJoCo's parent-child
    #  filter does not separate from them: contacts from the first step on, in the reference's model as here)
    g1 = s.get("con_geom1").astype(int)
    assert (g1 != 0).all()
    for _ in range(300):
        s.step()
    assert s.geti("fail") == 0 and np.isfinite(s.get("qpos")).all()
    g1, g2 = s.get("con_geom1").astype(int), s.get("con_geom2").astype(int)
    assert s.geti("ncon") >= 4 and any(m.geom_type[g] == GEOM_CAPSULE for g in g2[g1 == 0]), (g1, g2)
    f = s.get("efc_force")
    assert f.min() >= 0 and f.sum() > 0.5 * m.body_mass.sum() * 9.81  # it is carried by its contacts
    # the device library refuses what its kernels cannot hold, and says why
    import ctypes as C
    from uhc_amd._capi import model_desc
    from uhc_amd._lib import UhcError, check, lib
    h = C.c_void_p()
    with pytest.raises(UhcError, match="nv=147"):
        check(lib().uhc_model_create(C.byref(model_desc(m)), C.byref(h)))


def test_a_masterfoot_config_is_refused_by_the_env_with_the_reason():
    """`masterfoot: true` (config/smpl_shape/copycat_master_1.yml) must not run the plain humanoid silently: the env says what exists and what does not."""
    from types import SimpleNamespace
    from uhc_amd.envs.humanoid_im import VecHumanoidEnv
    cfg = SimpleNamespace(masterfoot=True, robot_cfg={"mesh": True})
    with pytest.raises(NotImplementedError, match="masterfoot.*nv <= 128"):
        VecHumanoidEnv(cfg, 4)
