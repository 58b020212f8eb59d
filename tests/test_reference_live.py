"""Live comparisons against the imported reference -- only where /root/reference exists (the build container; skipped on the GPU
box, which has no reference).  Nothing here is a GPU test; fixtures are not needed because both sides run in the same process."""
import glob
import json
import os
import sys
import tempfile

import numpy as np
import pytest

REF = os.environ.get("UHC_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "uhc")), reason="the reference only exists in the build container")

_PATHS = ("cfg_dict", "base_dir", "cfg_dir", "model_dir", "result_dir", "log_dir", "tb_dir", "output", "output_dir", "data_dir", "mujoco_model_file",
          "vis_model_file", "main_result_dir")


def _same(v, w):
    if isinstance(v, np.ndarray) or isinstance(w, np.ndarray):
        return np.array_equal(np.asarray(v, dtype=float), np.asarray(w, dtype=float), equal_nan=True)
    if isinstance(v, float) and np.isinf(v):
        return bool(np.isinf(w))
    return v == w


def test_config_equals_reference_on_every_shipped_config():
    """Every yml under the reference's config/ that the reference's own Config can load (85 of 115: the rest name model files it does
    not ship) gives the same attributes through uhc_amd's Config -- a yml written for the reference loads unchanged."""
    import yaml
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import ref_import
    import uhc  # noqa: F401  -- this build's alias package on purpose: the comparison must survive it being imported first
    with ref_import.reference_modules():
        from uhc.utils.config_utils.copycat_config import Config as RefConfig
        ref_import.assert_is_reference(RefConfig)
    from uhc_amd.utils.config_utils.copycat_config import Config as MyConfig
    assert RefConfig is not MyConfig
    assert os.path.realpath(RefConfig.__init__.__code__.co_filename).startswith(os.path.realpath(REF) + os.sep)
    assert os.sep + "uhc_amd" + os.sep in os.path.realpath(MyConfig.__init__.__code__.co_filename)
    assert sys.modules["uhc"] is uhc  # the alias is back for whoever runs after this test
    files = sorted(glob.glob(os.path.join(REF, "config", "**", "*.yml"), recursive=True))
    assert len(files) > 100
    compared = 0
    for f in files:
        cid = os.path.splitext(os.path.basename(f))[0]
        cd = yaml.safe_load(open(f))
        base = tempfile.mkdtemp()
        os.symlink(os.path.join(REF, "assets"), os.path.join(base, "assets"))
        try:
            rc = RefConfig(cfg_id=cid, base_dir=base, cfg_dict=json.loads(json.dumps(cd)))
        except OSError:
            continue  # the config names a model file the reference does not ship
        mc = MyConfig(cfg_id=cid, base_dir=tempfile.mkdtemp(), cfg_dict=json.loads(json.dumps(cd)))
        for k, v in vars(rc).items():
            if k in _PATHS:
                continue
            assert hasattr(mc, k), (f, k)
            assert _same(v, getattr(mc, k)), (f, k, v, getattr(mc, k))
        compared += 1
    assert compared >= 80


def test_new_boundary_helpers_equal_the_references():
    """get_qvel_fd_new, quat_mul_vec_batch, quaternion_from_euler_batch ('rzyx'), qpos_to_smpl, in_hull: this build's functions against the
    imported reference's on seeded inputs (both run here, in one process)."""
    from scipy.spatial import ConvexHull
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import ref_import
    with ref_import.reference_modules():
        import uhc.utils.math_utils as rmu
        import uhc.utils.transformation as rtf
        from uhc.smpllib import smpl_mujoco as rsm
        from uhc.smpllib.smpl_robot import in_hull as r_in_hull
        ref_import.assert_is_reference(rmu.get_qvel_fd_new)
        rng = np.random.default_rng(5)
        q0, q1 = rng.normal(size=76), rng.normal(size=76)
        for q in (q0, q1):
            q[3:7] /= np.linalg.norm(q[3:7])
        q1[7:12] += np.array([7.0, -7.0, 3.2, -3.2, 0.0])  # hinge differences beyond +-pi: the wrap
        ref_qvel = [rmu.get_qvel_fd_new(q0.copy(), q1.copy(), 1 / 30), rmu.get_qvel_fd_new(q0.copy(), q1.copy(), 1 / 30, "heading")]
        qs = rng.normal(size=(7, 4)); qs /= np.linalg.norm(qs, axis=1, keepdims=True)
        vs = rng.normal(size=(7, 3))
        ref_rot = rtf.quat_mul_vec_batch(qs.copy(), vs.copy())
        ang = rng.uniform(-3, 3, size=(3, 9))
        ref_eul = rtf.quaternion_from_euler_batch(ang[0].copy(), ang[1].copy(), ang[2].copy(), "rzyx")
        pts = rng.normal(size=(40, 3))
        hull = ConvexHull(pts)
        queries = rng.normal(scale=1.2, size=(200, 3))
        ref_in = r_in_hull(hull, queries)
        from uhc_amd.sim import load_asset_model
        model = load_asset_model()
        from uhc_amd.smpllib.smpl_mujoco import smpl_to_qpose
        qpos = smpl_to_qpose(rng.normal(scale=0.4, size=(5, 72)), model, trans=rng.normal(size=(5, 3)))
        ref_pose, ref_trans = rsm.qpos_to_smpl(qpos.copy(), model)
    from uhc_amd.smpllib.smpl_mujoco import qpos_to_smpl
    from uhc_amd.smpllib.smpl_robot import in_hull
    from uhc_amd.utils.math_utils import get_qvel_fd_new
    from uhc_amd.utils.transformation import quat_mul_vec_batch, quaternion_from_euler_batch
    np.testing.assert_allclose(get_qvel_fd_new(q0, q1, 1 / 30), ref_qvel[0], atol=1e-10)
    np.testing.assert_allclose(get_qvel_fd_new(q0, q1, 1 / 30, "heading"), ref_qvel[1], atol=1e-10)
    np.testing.assert_allclose(quat_mul_vec_batch(qs, vs), ref_rot, atol=1e-13)
    np.testing.assert_allclose(quaternion_from_euler_batch(ang[0], ang[1], ang[2], "rzyx"), ref_eul, atol=1e-13)
    np.testing.assert_array_equal(in_hull(hull, queries), ref_in)
    pose, trans = qpos_to_smpl(qpos, model)
    np.testing.assert_allclose(pose, ref_pose, atol=1e-10)
    np.testing.assert_allclose(trans, ref_trans, atol=1e-13)
