"""ctypes front-end of oracle/physics_oracle.c -- TEST INFRASTRUCTURE (checker only)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from uhc_amd._capi import UhcCtrlDesc, UhcModelDesc, ctrl_desc, model_desc

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libphysics_oracle.so")
        if not os.path.exists(so):
            from oracle.build import build
            build()
        L = C.CDLL(so)
        P = C.c_void_p
        L.orc_data_create.restype = P
        L.orc_data_create.argtypes = [C.POINTER(UhcModelDesc)]
        L.orc_data_free.argtypes = [P]
        for fn in ("orc_forward", "orc_step", "orc_euler", "orc_kinematics", "orc_com_pos", "orc_crb", "orc_factor_m",
                   "orc_collision", "orc_make_constraint", "orc_com_vel", "orc_passive", "orc_rne_bias",
                   "orc_fwd_acceleration", "orc_project_constraint", "orc_solve_pgs"):
            getattr(L, fn).argtypes = [C.POINTER(UhcModelDesc), P]
            getattr(L, fn).restype = None
        L.orc_set_state.argtypes = [C.POINTER(UhcModelDesc), P, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_do_simulation.argtypes = [C.POINTER(UhcModelDesc), C.POINTER(UhcCtrlDesc), P, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_do_simulation_mixed.argtypes = [C.POINTER(UhcModelDesc), C.POINTER(UhcCtrlDesc), P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_uint]
        L.orc_pd_torque.argtypes = [C.POINTER(UhcModelDesc), C.POINTER(UhcCtrlDesc), P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
        L.orc_rfc_implicit.argtypes = [C.POINTER(UhcModelDesc), C.POINTER(UhcCtrlDesc), P, C.POINTER(C.c_double)]
        L.orc_rfc_explicit.argtypes = [C.POINTER(UhcModelDesc), C.POINTER(UhcCtrlDesc), P, C.POINTER(C.c_double)]
        L.orc_batch_do_simulation.argtypes = [C.POINTER(UhcModelDesc), C.POINTER(UhcCtrlDesc), C.POINTER(P), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_get.argtypes = [C.POINTER(UhcModelDesc), P, C.c_char_p, C.POINTER(C.c_double), C.c_int]
        L.orc_get.restype = C.c_int
        L.orc_get_int.argtypes = [P, C.c_char_p]
        L.orc_get_int.restype = C.c_int
        L.orc_set.argtypes = [C.POINTER(UhcModelDesc), P, C.c_char_p, C.POINTER(C.c_double)]
        L.orc_full_m.argtypes = [C.POINTER(UhcModelDesc), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_solve_sparse.argtypes = [C.POINTER(UhcModelDesc), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleSim:
    """One environment of the CPU oracle."""

    def __init__(self, model, ctrl: UhcCtrlDesc | None = None):
        self.model = model
        self.desc = model_desc(model)
        self.ctrl = ctrl
        self.L = lib()
        self.d = self.L.orc_data_create(C.byref(self.desc))

    def __del__(self):
        try:
            self.L.orc_data_free(self.d)
        except Exception:
            pass

    def set_state(self, qpos, qvel):
        qpos = np.ascontiguousarray(qpos, dtype=np.float64)
        qvel = np.ascontiguousarray(qvel, dtype=np.float64)
        self.L.orc_set_state(C.byref(self.desc), self.d, _dp(qpos), _dp(qvel))

    def call(self, stage: str):
        getattr(self.L, "orc_" + stage)(C.byref(self.desc), self.d)

    def forward(self):
        self.call("forward")

    def step(self):
        self.call("step")

    def do_simulation(self, action, target_base, redo=0):
        """redo: the device path's UHC_F_REDO word of the same step; its bits 8+ name the substeps the device solved by sweeps after its
        exact solve gave up, and the checker takes solver 0 in exactly those."""
        action = np.ascontiguousarray(action, dtype=np.float64)
        target_base = np.ascontiguousarray(target_base, dtype=np.float64)
        self.L.orc_do_simulation_mixed(C.byref(self.desc), C.byref(self.ctrl), self.d, _dp(action), _dp(target_base), (int(redo) >> 8) & 0x1fffff)

    def pd_torque(self, action, target_base, it):
        action = np.ascontiguousarray(action, dtype=np.float64)
        target_base = np.ascontiguousarray(target_base, dtype=np.float64)
        self.L.orc_pd_torque(C.byref(self.desc), C.byref(self.ctrl), self.d, _dp(action), _dp(target_base), int(it))
        return self.get("ctrl")

    def rfc_implicit(self, action):
        action = np.ascontiguousarray(action, dtype=np.float64)
        self.L.orc_rfc_implicit(C.byref(self.desc), C.byref(self.ctrl), self.d, _dp(action))
        return self.get("qfrc_applied")

    def rfc_explicit(self, action):
        action = np.ascontiguousarray(action, dtype=np.float64)
        self.L.orc_rfc_explicit(C.byref(self.desc), C.byref(self.ctrl), self.d, _dp(action))
        return self.get("qfrc_applied")

    def get(self, name: str) -> np.ndarray:
        n = self.L.orc_get(C.byref(self.desc), self.d, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.zeros(max(n, 1))
        self.L.orc_get(C.byref(self.desc), self.d, name.encode(), _dp(out), n)
        return out[:n]

    def geti(self, name: str) -> int:
        return self.L.orc_get_int(self.d, name.encode())

    def set(self, name: str, val):
        val = np.ascontiguousarray(val, dtype=np.float64)
        self.L.orc_set(C.byref(self.desc), self.d, name.encode(), _dp(val))

    def full_m(self) -> np.ndarray:
        qM = self.get("qM")
        out = np.zeros((self.model.nv, self.model.nv))
        self.L.orc_full_m(C.byref(self.desc), _dp(qM), _dp(out))
        return out
