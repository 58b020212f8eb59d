"""Host-side pieces of the env layer's objects (no GPU): which bodies count as objects, the synthetic obj_pose clips, the loader's
has_obj / num_obj keys (uhc/data_loaders/dataset_amass_single.py:246-247), the descriptor the C-ABI takes."""
import numpy as np


def test_trailing_free_bodies_and_object_parameters(model):
    from uhc_amd.model.mjcf import add_free_bodies, ball_variant, trailing_free_bodies
    from uhc_amd.model.shapes import box_triangles
    assert trailing_free_bodies(model) == 0 and trailing_free_bodies(ball_variant(model)) == 0  # the humanoid's own free root is not an object
    poses = np.array([[1.0, 0, 0.5, 1, 0, 0, 0], [0, 1.0, 0.5, 0.6, 0.8, 0, 0], [2.0, 0, 0.5, 1, 0, 0, 0]])
    m = add_free_bodies(model, [box_triangles(0.1, 0.1, 0.1)] * 3, poses, density=1000.0, friction=1.0, condim=3)
    assert trailing_free_bodies(m) == 3 and (m.nq, m.nv, m.nbody) == (model.nq + 21, model.nv + 18, model.nbody + 3)
    np.testing.assert_allclose(m.geom_friction[-3:], [[1.0, 0.005, 0.0001]] * 3)  # uhc/smpllib/smpl_robot.py:1216-1224
    assert (m.geom_condim[-3:] == 3).all() and (m.geom_contype[-3:] == 1).all() and (m.geom_conaffinity[-3:] == 1).all()
    np.testing.assert_allclose(m.body_mass[-3:], 1000.0 * 0.008, rtol=1e-12)
    np.testing.assert_allclose(m.qpos0[model.nq:].reshape(3, 7), poses / np.r_[np.ones(3), np.ones(4)], atol=1e-15)
    # the humanoid in front of the objects is untouched: the reference's qpos_lim / qvel_lim / body_lim
    np.testing.assert_array_equal(m.qpos0[:model.nq], model.qpos0)
    assert m.body_names[:model.nbody] == model.body_names


def test_synthetic_clips_with_objects_feed_the_loader():
    from uhc_amd.data_loaders.dataset_amass_single import DatasetAMASSSingle
    from uhc_amd.data_loaders.synthetic import make_synthetic_amass
    clips = make_synthetic_amass(3, seed=2, t_range=(40, 60), objects=4)
    for c in clips.values():
        T = c["pose_aa"].shape[0]
        assert c["obj_pose"].shape == (T, 28)
        op = c["obj_pose"].reshape(T, 4, 7)
        np.testing.assert_allclose(np.linalg.norm(op[..., 3:], axis=-1), 1.0)
        assert (op[..., 2] > 0.15).all()  # above the floor: they are dropped
    dl = DatasetAMASSSingle({"file_path": "synthetic", "t_min": 15, "t_max": 30, "fr_num": 90}, "train", pickle_data=clips)
    s = dl.get_sample_from_key(dl.data_keys[0], full_sample=False, fr_start=3)
    assert s["has_obj"] and s["num_obj"] == 4 and s["obj_pose"].shape == (s["pose_aa"].shape[0], 28)
    plain = DatasetAMASSSingle({"file_path": "synthetic", "t_min": 15, "t_max": 30, "fr_num": 90}, "train", pickle_data=make_synthetic_amass(2, seed=2, t_range=(40, 60)))
    s = plain.get_sample_from_key(plain.data_keys[0], full_sample=True)
    assert not s["has_obj"] and s["num_obj"] == 0


def test_env_descriptor_carries_num_obj(model):
    from uhc_amd._capi import UhcEnvDesc, env_desc
    d = env_desc(model, num_obj=3)
    assert d.num_obj == 3 and env_desc(model).num_obj == 0
    assert [n for n, _ in UhcEnvDesc._fields_][-1] == "num_obj"  # appended: ABI 8 (include/uhc_amd.h)
