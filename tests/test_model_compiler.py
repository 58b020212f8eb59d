"""Host model compiler (uhc_amd/model/mjcf.py): known-answer tests + the committed asset."""
import os

import numpy as np
import pytest

from tests.helpers import box_model, box_triangles, pendulum_model
from uhc_amd.model import mjcf

REF_XML = "/root/reference/assets/mujoco_models/humanoid_smpl_neutral_mesh.xml"


def test_box_mesh_mass_properties():
    tris = box_triangles(0.1, 0.2, 0.3, center=(1.0, -2.0, 0.5))
    v, f = mjcf.weld(tris)
    assert v.shape == (8, 3)
    vol, com, I = mjcf.polyhedron_mass_properties(v, f)
    assert vol == pytest.approx(8 * 0.1 * 0.2 * 0.3, rel=1e-12)
    np.testing.assert_allclose(com, [1.0, -2.0, 0.5], atol=1e-12)
    m = vol
    expect = m / 3 * np.diag([0.2 ** 2 + 0.3 ** 2, 0.1 ** 2 + 0.3 ** 2, 0.1 ** 2 + 0.2 ** 2])
    np.testing.assert_allclose(I, expect, atol=1e-12)


def test_tetrahedron_volume_and_com():
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]])
    f = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]])
    vol, com, I = mjcf.polyhedron_mass_properties(v, f)
    assert vol == pytest.approx(1 / 6)
    np.testing.assert_allclose(com, [0.25, 0.25, 0.25], atol=1e-12)
    # Ixx of the unit right tetrahedron about its COM, unit density: 3/80 * vol * ... known value 1/80*... use integral check
    # integral (y^2+z^2) dV about origin = 2 * (1/60) ; shift to COM
    Ixx0 = 2.0 / 60.0
    assert I[0, 0] == pytest.approx(Ixx0 - vol * (0.25 ** 2 + 0.25 ** 2), rel=1e-12)


def test_pendulum_model_tables():
    m = pendulum_model()
    assert (m.nq, m.nv, m.nu, m.nbody, m.njnt) == (1, 1, 1, 2, 1)
    assert m.body_mass[1] == pytest.approx(1000 * 0.1 ** 3)
    np.testing.assert_allclose(m.body_ipos[1], [0, 0, -0.5], atol=1e-12)
    assert m.dof_parentid[0] == -1 and m.dof_madr.tolist() == [0, 1]
    # inertia about the hinge: I_com + m l^2 ; dof_invweight0 = 1/M
    I = m.body_mass[1] * (0.1 ** 2 + 0.1 ** 2) / 12 + m.body_mass[1] * 0.25
    assert m.dof_invweight0[0] == pytest.approx(1 / I, rel=1e-10)


def test_free_body_constants():
    m = box_model(0.1)
    assert (m.nq, m.nv, m.ngeom) == (7, 6, 2)
    mass = 1000 * 0.2 ** 3
    assert m.body_mass[1] == pytest.approx(mass)
    # free body: translational invweight = 1/m, rotational = 1/I
    I = mass * (0.2 ** 2 + 0.2 ** 2) / 12
    np.testing.assert_allclose(m.dof_invweight0, [1 / mass] * 3 + [1 / I] * 3, rtol=1e-9)
    np.testing.assert_allclose(m.body_invweight0[1], [1 / mass, 1 / I], rtol=1e-9)
    np.testing.assert_allclose(m.qpos0, [0, 0, 0.1, 1, 0, 0, 0])
    # hull graph: every box vertex has 3 axis neighbours + diagonals from the triangulation
    deg = np.diff(m.mesh_adjadr)
    assert deg.min() >= 3 and deg.sum() == m.nmeshadj


def test_asset_model_layout():
    from uhc_amd.sim import load_asset_model
    m = load_asset_model()
    assert (m.nq, m.nv, m.nu, m.nbody, m.njnt, m.ngeom, m.nM) == (76, 75, 69, 25, 70, 25, 1221)
    assert m.body_names[1] == "Pelvis" and m.body_names[-1] == "R_Hand"
    assert m.timestep == pytest.approx(0.00222222222)
    # longest dof chain Pelvis..Hand = 30
    depth = np.zeros(m.nv, dtype=int)
    for i in range(m.nv):
        depth[i] = 0 if m.dof_parentid[i] < 0 else depth[m.dof_parentid[i]] + 1
    assert depth.max() + 1 == 30
    # floor collides with every body geom, body geoms do not collide with each other
    assert m.geom_type[0] == mjcf.GEOM_PLANE and (m.geom_contype[1:] == 0).all() and (m.geom_conaffinity == 1).all()


@pytest.mark.skipif(not os.path.exists(REF_XML), reason="reference assets only exist in the build container")
def test_asset_matches_fresh_compile_of_reference_xml():
    from uhc_amd.sim import load_asset_model
    a = load_asset_model()
    b = mjcf.compile_mjcf_file(REF_XML)
    for name in ("body_pos", "body_mass", "body_inertia", "body_ipos", "mesh_vert", "dof_invweight0", "body_invweight0", "jnt_range"):
        np.testing.assert_allclose(getattr(a, name), getattr(b, name), rtol=0, atol=1e-12, err_msg=name)
    assert a.body_names == b.body_names and a.actuator_names == b.actuator_names


def test_shape_to_model_generator_round_trip(model):
    """uhc_amd/smpllib/smpl_robot.py (the reference's Robot.load_from_skeleton pipeline): fed with the neutral asset's own hull vertices as
    'SMPL vertices' (one-hot skin weights, joints = body origins), the generator must give the asset back -- same tree and names, same bone
    offsets, the same masses up to the vertices the 50-vertex budget removes -- with body-body collisions and the two excludes switched on."""
    import numpy as np
    from uhc_amd.model.mjcf import kinematics_np, quat_to_mat
    from uhc_amd.smpllib.smpl_mujoco import SMPL_BONE_ORDER_NAMES
    from uhc_amd.smpllib.smpl_robot import Robot, SMPLBody, decimate_hull
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
    for ball in (False, True):
        r = Robot({"mesh": True, "ball": ball}, body_provider=lambda b, g: (verts, joints, W))
        r.load_from_skeleton(np.zeros(16), gender=[0])
        assert b"<mujoco" in r.export_xml_string()
        g = r.get_model()
        assert g.body_names == model.body_names and g.nbody == 25 and g.nv == 75 and g.nu == 69 and g.nq == (99 if ball else 76)
        np.testing.assert_allclose(g.body_pos, model.body_pos, atol=1e-4)  # the XML carries 4 decimals, like the reference's writer
        np.testing.assert_allclose(g.body_mass, model.body_mass, rtol=0.02)
        assert (g.geom_contype[1:] == 1).all() and g.nexclude == 2 and g.geom_vertnum.max() <= 50
        assert g.actuator_names == model.actuator_names
        if not ball:  # rel_joint_lm knee range (smpl_robot.py:1087-1092)
            j = g.joint_names.index("L_Knee_x")
            np.testing.assert_allclose(g.jnt_range[j], [-np.pi / 16, np.pi], atol=1e-4)
    # the vertex budget: a dense ellipsoid comes down to 50 hull vertices and keeps most of its volume
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(0)
    P = rng.normal(size=(800, 3))
    P = P / np.linalg.norm(P, axis=1)[:, None] * [0.1, 0.05, 0.2]
    D = decimate_hull(P, 50)
    assert len(D) == 50 and ConvexHull(D).volume > 0.85 * ConvexHull(P).volume
    # the SMPL forward pass itself needs the licensed model files
    import pytest
    with pytest.raises(FileNotFoundError):
        SMPLBody("/nonexistent")(np.zeros(10), 0)
