"""ctypes mirror of include/uhc_amd.h (structs only; the loader lives in uhc_amd/_lib.py)."""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np

_I32P = C.POINTER(C.c_int32)
_F64P = C.POINTER(C.c_double)

_MODEL_INT_SCALARS = ["nq", "nv", "nu", "nbody", "njnt", "ngeom", "nmeshvert", "nmeshadj", "nexclude",
                      "iterations", "plane_mesh_maxcon", "solver"]
_MODEL_PTRS = [
    ("body_parentid", "i"), ("body_jntadr", "i"), ("body_jntnum", "i"), ("body_dofadr", "i"), ("body_dofnum", "i"),
    ("body_pos", "d"), ("body_quat", "d"), ("body_ipos", "d"), ("body_iquat", "d"),
    ("body_mass", "d"), ("body_inertia", "d"), ("body_invweight0", "d"),
    ("jnt_type", "i"), ("jnt_bodyid", "i"), ("jnt_qposadr", "i"), ("jnt_dofadr", "i"), ("jnt_limited", "i"),
    ("jnt_pos", "d"), ("jnt_axis", "d"), ("jnt_range", "d"), ("jnt_stiffness", "d"), ("jnt_margin", "d"),
    ("qpos0", "d"), ("qpos_spring", "d"),
    ("dof_bodyid", "i"), ("dof_jntid", "i"), ("dof_parentid", "i"), ("dof_madr", "i"),
    ("dof_armature", "d"), ("dof_damping", "d"), ("dof_frictionloss", "d"), ("dof_invweight0", "d"),
    ("geom_type", "i"), ("geom_bodyid", "i"), ("geom_contype", "i"), ("geom_conaffinity", "i"), ("geom_condim", "i"),
    ("geom_vertadr", "i"), ("geom_vertnum", "i"),
    ("geom_pos", "d"), ("geom_quat", "d"), ("geom_size", "d"), ("geom_friction", "d"),
    ("geom_margin", "d"), ("geom_gap", "d"), ("geom_solref", "d"), ("geom_solimp", "d"),
    ("geom_rbound", "d"), ("geom_center", "d"),
    ("mesh_vert", "d"), ("mesh_adjadr", "i"), ("mesh_adj", "i"), ("exclude_pair", "i"),
    ("actuator_dofid", "i"), ("actuator_gear", "d"),
]


class UhcModelDesc(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in _MODEL_INT_SCALARS]
                + [("timestep", C.c_double), ("tolerance", C.c_double), ("meaninertia", C.c_double),
                   ("gravity", C.c_double * 3)]
                + [(n, _I32P if t == "i" else _F64P) for n, t in _MODEL_PTRS])


class UhcCtrlDesc(C.Structure):
    _fields_ = [("n_substeps", C.c_int32), ("action_type", C.c_int32), ("meta_pd", C.c_int32),
                ("rfc_mode", C.c_int32), ("action_dim", C.c_int32), ("body_vf_dim", C.c_int32),
                ("rfc_scale", C.c_double), ("rfc_lim", C.c_double), ("base_rot", C.c_double * 4),
                ("jkp", _F64P), ("jkd", _F64P), ("torque_lim", _F64P), ("a_scale", _F64P),
                ("vf_body", _I32P), ("n_vf_body", C.c_int32), ("_pad", C.c_int32)]


def model_desc(model) -> UhcModelDesc:
    """Build a UhcModelDesc whose pointers reference contiguous copies kept alive on the struct."""
    d = UhcModelDesc()
    keep: List[np.ndarray] = []
    for n in _MODEL_INT_SCALARS:
        setattr(d, n, int(getattr(model, n, 0)))
    d.timestep, d.tolerance, d.meaninertia = float(model.timestep), float(model.tolerance), float(model.meaninertia)
    d.gravity = (C.c_double * 3)(*[float(x) for x in model.gravity])
    for n, t in _MODEL_PTRS:
        arr = np.ascontiguousarray(getattr(model, n), dtype=np.int32 if t == "i" else np.float64)
        if n == "actuator_gear" and arr.ndim == 1:  # scalar gears of hinge models -> [nu][3]
            arr = np.ascontiguousarray(np.stack([arr, np.zeros_like(arr), np.zeros_like(arr)], axis=1))
        if arr.size == 0:
            arr = np.zeros(1, dtype=arr.dtype)
        keep.append(arr)
        setattr(d, n, arr.ctypes.data_as(_I32P if t == "i" else _F64P))
    d._keep = keep
    return d


def ctrl_desc(n_substeps=15, action_type=0, meta_pd=0, rfc_mode=0, action_dim=0, rfc_scale=0.0, rfc_lim=100.0,
              base_rot=(0.7071, 0.7071, 0.0, 0.0), jkp=None, jkd=None, torque_lim=None, a_scale=None, vf_body=None,
              body_vf_dim=9) -> UhcCtrlDesc:
    c = UhcCtrlDesc()
    c.n_substeps, c.action_type, c.meta_pd, c.rfc_mode, c.action_dim = n_substeps, action_type, meta_pd, rfc_mode, action_dim
    c.rfc_scale, c.rfc_lim = float(rfc_scale), float(rfc_lim)
    c.base_rot = (C.c_double * 4)(*[float(x) for x in base_rot])
    keep = []
    for n, v in (("jkp", jkp), ("jkd", jkd), ("torque_lim", torque_lim), ("a_scale", a_scale)):
        arr = np.ascontiguousarray(v if v is not None else np.zeros(1), dtype=np.float64)
        keep.append(arr)
        setattr(c, n, arr.ctypes.data_as(_F64P))
    vb = np.ascontiguousarray(vf_body if vf_body is not None else np.zeros(1), dtype=np.int32)
    keep.append(vb)
    c.vf_body, c.n_vf_body, c.body_vf_dim = vb.ctypes.data_as(_I32P), (len(vf_body) if vf_body is not None else 0), int(body_vf_dim)
    c._keep = keep
    return c


class UhcEnvDesc(C.Structure):
    _fields_ = [("obs_v", C.c_int32), ("has_shape", C.c_int32), ("env_episode_len", C.c_int32),
                ("env_expert_trail_steps", C.c_int32), ("ee_body", C.c_int32 * 5), ("reward_v", C.c_int32),
                ("body_diff_thresh", C.c_double), ("reward_weights", C.c_double * 16), ("jpos_diffw", _F64P),
                ("fut_frames", C.c_int32), ("fut_skip", C.c_int32), ("obs_flags", C.c_int32), ("reward_jpos_diffw", _F64P), ("term_body", C.c_int32), ("num_obj", C.c_int32)]


REWARD_KEYS = ("w_p", "w_v", "w_e", "w_c", "w_vf", "k_p", "k_v", "k_e", "k_c", "k_vf", "w_wp", "w_j", "k_wp", "k_j")
REWARD_DEFAULTS = (0.6, 0.1, 0.2, 0.1, 0.0, 2.0, 0.005, 20.0, 1000.0, 1.0, 0.0, 0.0, 0.0, 0.0)  # reward_function.py:16-29
# world_rfc_implicit_v2 / v3 read their own defaults (reward_function.py:646-654, :729-747)
REWARD_DEFAULTS_V23 = dict(w_p=0.4, w_wp=0.4, w_v=0.005, w_j=100.0, w_c=100.0, w_vf=1.0, k_p=0.4, k_wp=0.4, k_v=0.005, k_j=100.0, k_c=100.0, k_vf=1.0,
                           w_e=0.0, k_e=0.0)
REWARD_IDS = {"world_rfc_implicit": 0, "world_rfc_implicit_quat": 0, "world_rfc_explicit": 1, "world_rfc_implicit_v1_mul": 2,
              "world_rfc_explicit_mul": 3, "world_rfc_implicit_v2": 4, "world_rfc_implicit_v3": 5}
REWARD_PARTS = {0: 5, 1: 5, 2: 5, 3: 5, 4: 6, 5: 6}


def env_desc(model, *, obs_v=2, has_shape=True, env_episode_len=100000, env_expert_trail_steps=0, body_diff_thresh=0.5,
             reward_weights=None, jpos_diffw=None, reward_v=0, fut_frames=10, fut_skip=10, obs_heading=False, root_deheading=False,
             obs_phase=True, obs_vel="full", env_term_body="body", num_obj=0) -> UhcEnvDesc:
    """`model`: the HUMANOID's model (body names, weights); num_obj: free objects behind it in the batch's model (expert["num_obj"])."""
    from .smpllib.smpl_mujoco import SMPL_EE_NAMES, SMPLConverter
    d = UhcEnvDesc()
    d.obs_v, d.has_shape, d.env_episode_len, d.env_expert_trail_steps = obs_v, int(has_shape), int(env_episode_len), int(env_expert_trail_steps)
    d.ee_body = (C.c_int32 * 5)(*[model.body_names.index(n) for n in SMPL_EE_NAMES])
    d.body_diff_thresh = float(body_diff_thresh)
    d.reward_v = int(reward_v)
    d.fut_frames, d.fut_skip = int(fut_frames), int(fut_skip)
    d.term_body = {"body": 0, "root": 1}[env_term_body]
    d.num_obj = int(num_obj)
    d.obs_flags = int(bool(obs_heading)) | int(bool(root_deheading)) << 1 | int(bool(obs_phase)) << 2 | int(obs_vel == "root") << 3
    rw = dict(REWARD_DEFAULTS_V23) if reward_v >= 4 else dict(zip(REWARD_KEYS, REWARD_DEFAULTS))
    given = dict(reward_weights or {})
    rjw = given.pop("jpos_diffw", None)
    rw.update(given)
    d.reward_weights = (C.c_double * 16)(*[float(rw[k]) for k in REWARD_KEYS], 0.0, 0.0)
    w = np.ascontiguousarray(jpos_diffw if jpos_diffw is not None else SMPLConverter(model, model).get_new_diff_weight(), dtype=np.float64)
    d.jpos_diffw = w.ctypes.data_as(_F64P)
    d._keep = [w]
    if rjw is not None:
        rjw = np.ascontiguousarray(rjw, dtype=np.float64)
        assert rjw.shape == (model.nbody - 1,), "reward_weights['jpos_diffw'] needs one weight per body"
        d._keep.append(rjw)
        d.reward_jpos_diffw = rjw.ctypes.data_as(_F64P)
    return d
