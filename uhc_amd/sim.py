"""Batched simulation handle over the C-ABI (include/uhc_amd.h).  torch is used only for device
memory and streams; every computation happens inside libuhc_amd.so."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from ._capi import UhcCtrlDesc, model_desc
from ._lib import check, lib

F_QPOS, F_QVEL, F_XPOS, F_XQUAT, F_XIPOS, F_QM, F_QFRC_BIAS, F_QACC, F_CTRL = range(9)
F_NCON, F_NEFC, F_FAIL, F_SOLVER_ITER, F_QFRC_APPLIED, F_EFC_OVERFLOW, F_STAGE_PROF = range(9, 16)
_INT_FIELDS = {F_NCON, F_NEFC, F_FAIL, F_SOLVER_ITER, F_EFC_OVERFLOW}


class _DevView:
    """__cuda_array_interface__ view of library-owned device memory (no copy, no ownership)."""

    def __init__(self, ptr: int, shape, typestr: str, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}
        self._owner = owner


class SimBatch:
    def __init__(self, models, ctrl: UhcCtrlDesc, n_env: int, env_model: Optional[Sequence[int]] = None, device: int = 0):
        if not isinstance(models, (list, tuple)):
            models = [models]
        if not torch.cuda.is_available():
            raise RuntimeError("uhc_amd.SimBatch needs an MI355X (torch.cuda unavailable); there is no CPU fallback")
        self.L = lib()
        self.models = list(models)
        self.model = self.models[0]
        self.n_env = int(n_env)
        self.device = torch.device("cuda", device)
        self.ctrl = ctrl
        self._descs = [model_desc(m) for m in self.models]
        self._mh = []
        for d in self._descs:
            h = C.c_void_p()
            check(self.L.uhc_model_create(C.byref(d), C.byref(h)))
            self._mh.append(h)
        arr = (C.c_void_p * len(self._mh))(*[h.value for h in self._mh])
        em = None
        if env_model is not None:
            em_np = np.ascontiguousarray(env_model, dtype=np.int32)
            assert em_np.shape == (n_env,)
            em = em_np.ctypes.data_as(C.POINTER(C.c_int32))
        self._b = C.c_void_p()
        check(self.L.uhc_batch_create(arr, len(self._mh), em, self.n_env, device, C.byref(ctrl), C.byref(self._b)))
        self.nM = self.L.uhc_model_nM(self._mh[0])
        self._fields = {}
        self.use_current_stream()

    def close(self):
        if getattr(self, "_b", None) is not None and self._b:
            self.L.uhc_batch_free(self._b)
            self._b = None
            for h in self._mh:
                self.L.uhc_model_free(h)
            self._mh = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- zero-copy torch views of the library-owned state
    def field(self, f: int) -> torch.Tensor:
        if f in self._fields:
            return self._fields[f]
        p, n = C.c_void_p(), C.c_int64()
        check(self.L.uhc_batch_field(self._b, f, C.byref(p), C.byref(n)))
        per = n.value // self.n_env
        if f == F_STAGE_PROF:
            t = torch.as_tensor(_DevView(p.value, (self.n_env, 16), "<i8", self), device=self.device)
        elif f in _INT_FIELDS:
            t = torch.as_tensor(_DevView(p.value, (self.n_env,), "<i4", self), device=self.device)
        else:
            t = torch.as_tensor(_DevView(p.value, (self.n_env, per), "<f8", self), device=self.device)
        self._fields[f] = t
        return t

    def use_current_stream(self):
        check(self.L.uhc_batch_set_stream(self._b, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def sync(self):
        check(self.L.uhc_batch_sync(self._b))

    def set_rfc_scale(self, s: float):
        check(self.L.uhc_batch_set_rfc_scale(self._b, float(s)))

    def set_state(self, qpos: torch.Tensor, qvel: torch.Tensor, env_ids: Optional[torch.Tensor] = None):
        qpos = qpos.to(self.device, torch.float64).contiguous()
        qvel = qvel.to(self.device, torch.float64).contiguous()
        n = qpos.shape[0]
        ids = None
        if env_ids is not None:
            env_ids = env_ids.to(self.device, torch.int32).contiguous()
            assert env_ids.shape[0] == n
            ids = C.c_void_p(env_ids.data_ptr())
        check(self.L.uhc_batch_set_state(self._b, ids, n, C.c_void_p(qpos.data_ptr()), C.c_void_p(qvel.data_ptr())))
        self._keep = (qpos, qvel, env_ids)

    def simulate(self, action: torch.Tensor, target_base: torch.Tensor, active: Optional[torch.Tensor] = None):
        assert action.dtype == torch.float64 and action.is_contiguous() and action.shape == (self.n_env, self.ctrl.action_dim)
        assert target_base.dtype == torch.float64 and target_base.is_contiguous() and target_base.shape == (self.n_env, self.model.nu)
        a = None
        if active is not None:
            assert active.dtype == torch.int32 and active.is_contiguous()
            a = C.c_void_p(active.data_ptr())
        check(self.L.uhc_batch_simulate(self._b, C.c_void_p(action.data_ptr()), C.c_void_p(target_base.data_ptr()), a))

    def forward(self):
        check(self.L.uhc_batch_forward(self._b))


def make_ctrl(model, *, meta_pd: bool = True, meta_pd_joint: bool = False, residual_force: bool = True,
              residual_force_mode: str = "implicit", residual_force_scale: float = 100.0, residual_force_lim: float = 100.0,
              rfc_rate: float = 1.0, action_type: str = "position", pd_mul: float = 1.0, tq_mul: float = 1.0,
              base_rot=(0.7071, 0.7071, 0.0, 0.0), n_substeps: int = 15) -> UhcCtrlDesc:
    """UhcCtrlDesc from the reference's config knobs (copycat_config.py:86-113, humanoid_im.py:120-124,226-255)."""
    from ._capi import ctrl_desc
    from .smpllib.smpl_mujoco import SMPLConverter

    conv = SMPLConverter(model, model)
    nu = model.nu
    rfc_mode = 0
    vf_dim = 0
    if residual_force:
        if residual_force_mode != "implicit":
            raise NotImplementedError("explicit RFC is a later row (SURVEY.md 8f-4)")
        rfc_mode, vf_dim = 1, 6
    mp = 1 if meta_pd else (2 if meta_pd_joint else 0)
    mp_dim = 2 * n_substeps if mp == 1 else (2 * nu if mp == 2 else 0)
    return ctrl_desc(n_substeps=n_substeps, action_type=0 if action_type == "position" else 1, meta_pd=mp, rfc_mode=rfc_mode,
                     action_dim=nu + vf_dim + mp_dim, rfc_scale=residual_force_scale * rfc_rate, rfc_lim=residual_force_lim,
                     base_rot=base_rot, jkp=conv.get_new_jkp() * pd_mul, jkd=conv.get_new_jkd() * pd_mul,
                     torque_lim=conv.get_new_torque_limit() * tq_mul, a_scale=conv.get_new_a_scale())


def load_asset_model(name: str = "humanoid_smpl_neutral_mesh"):
    import os
    from .model.mjcf import Model
    return Model.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", name + ".npz"))
