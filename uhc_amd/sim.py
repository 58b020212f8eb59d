"""Batched simulation handle over the C-ABI (include/uhc_amd.h).  torch is used only for device
memory and streams; every computation happens inside libuhc_amd.so."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np
import torch

from ._capi import UhcCtrlDesc, UhcEnvDesc, model_desc
from ._lib import check, lib

F_QPOS, F_QVEL, F_XPOS, F_XQUAT, F_XIPOS, F_QM, F_QFRC_BIAS, F_QACC, F_CTRL = range(9)
F_NCON, F_NEFC, F_FAIL, F_SOLVER_ITER, F_QFRC_APPLIED, F_EFC_OVERFLOW, F_STAGE_PROF, F_REDO, F_TIER, F_HANDON_WHY = range(9, 19)
_INT_FIELDS = {F_NCON, F_NEFC, F_FAIL, F_SOLVER_ITER, F_EFC_OVERFLOW, F_REDO, F_TIER, F_HANDON_WHY}


class _DevView:
    """__cuda_array_interface__ view of library-owned device memory (no copy, no ownership)."""

    def __init__(self, ptr: int, shape, typestr: str, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}
        self._owner = owner


_DEFERRED = []  # handles whose owner was finalised while a stream was capturing: freed by the next close() / SimBatch creation outside a capture


def _capturing() -> bool:
    try:
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    except Exception:
        return False


def _drain_deferred() -> None:
    if not _DEFERRED or _capturing():
        return
    pending, _DEFERRED[:] = list(_DEFERRED), []
    for L, kind, h, models in sorted(pending, key=lambda x: x[1] != "env"):  # env layers before the batches they sit on
        if kind == "env":
            L.uhc_env_free(h)
        else:
            L.uhc_batch_free(h)
            for mh in models:
                L.uhc_model_free(mh)


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
        _drain_deferred()
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
            if _capturing():  # (a finaliser that fires inside a stream capture: hipFree is not allowed there -- the handles wait for the next close / creation)
                _DEFERRED.append((self.L, "batch", self._b, list(self._mh)))
                self._b, self._mh = None, []
                return
            _drain_deferred()
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
            t = torch.as_tensor(_DevView(p.value, (self.n_env, 40), "<i8", self), device=self.device)
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

    def set_overflow_mode(self, truncate: bool):
        """False (default): envs beyond the fast kernel's 16 contacts / 64 rows are redone exactly by the general kernel;
        True: they keep what fits (reported in F_EFC_OVERFLOW) and no second pass is needed."""
        check(self.L.uhc_batch_set_overflow_mode(self._b, int(bool(truncate))))

    def set_kernel_path(self, mode):
        """0 / False: tier chain (fast kernel, then the general tier on the envs beyond its capacity, then the large tier); 1 / True: general
        tier first (scenes where most envs exceed the fast kernel's 64 rows); 2: sticky tiers -- every env starts in the tier that computed
        its last step, the slower tiers run beside the fast one on a side stream (include/uhc_amd.h)."""
        check(self.L.uhc_batch_set_kernel_path(self._b, int(mode)))

    def set_solver(self, solver: int, iterations: int = 0):
        """Contact solver of the following launches: 0 = PGS sweeps (cap `iterations`), 1 = exact active-set solve."""
        check(self.L.uhc_batch_set_solver(self._b, int(solver), int(iterations)))

    def set_rfc_scale(self, s: float):
        check(self.L.uhc_batch_set_rfc_scale(self._b, float(s)))

    def set_state(self, qpos: torch.Tensor, qvel: torch.Tensor, env_ids: Optional[torch.Tensor] = None):
        qpos = qpos.to(self.device, torch.float64).contiguous()
        qvel = qvel.to(self.device, torch.float64).contiguous()
        n = qpos.shape[0]
        # the library reads n x nq and n x nv doubles: a pose of another model's width (a hinge pose handed to the ball-joint model) is a read
        # past the tensor and a humanoid put together from whatever lies there
        if qpos.ndim != 2 or qvel.ndim != 2 or qpos.shape[1] != self.model.nq or qvel.shape != (n, self.model.nv):
            raise ValueError(f"set_state: qpos {tuple(qpos.shape)} / qvel {tuple(qvel.shape)} for a model with nq {self.model.nq}, nv {self.model.nv}")
        ids = None
        if env_ids is not None:
            env_ids = env_ids.to(self.device, torch.int32).contiguous()
            assert env_ids.shape[0] == n
            ids = C.c_void_p(env_ids.data_ptr())
        check(self.L.uhc_batch_set_state(self._b, ids, n, C.c_void_p(qpos.data_ptr()), C.c_void_p(qvel.data_ptr())))
        self._keep = (qpos, qvel, env_ids)

    def simulate(self, action: torch.Tensor, target_base: torch.Tensor, active: Optional[torch.Tensor] = None):
        assert action.dtype == torch.float64 and action.is_contiguous() and action.shape == (self.n_env, self.ctrl.action_dim)
        if self.model.nu == 0:  # unactuated model: the library still wants non-null buffers
            target_base = action
        assert target_base.dtype == torch.float64 and target_base.is_contiguous() and (self.model.nu == 0 or target_base.shape == (self.n_env, self.model.nu))
        a = None
        if active is not None:
            assert active.dtype == torch.int32 and active.is_contiguous()
            a = C.c_void_p(active.data_ptr())
        check(self.L.uhc_batch_simulate(self._b, C.c_void_p(action.data_ptr()), C.c_void_p(target_base.data_ptr()), a))

    def forward(self):
        check(self.L.uhc_batch_forward(self._b))

    def set_timing(self, enable: bool):
        check(self.L.uhc_batch_set_timing(self._b, int(enable)))

    def kernel_time(self):
        """(total ms, launches) of the fused step kernel since the last call (HIP events on the batch's stream)."""
        ms, n = C.c_double(), C.c_int32()
        check(self.L.uhc_batch_kernel_time(self._b, C.byref(ms), C.byref(n)))
        return ms.value, n.value


E_OBS, E_REWARD, E_REWARD_PARTS, E_DONE, E_FAIL, E_END, E_PERCENT, E_CUR_T, E_BODY_DIFF, E_TARGET_BASE, E_CONSUMED, E_EPISODE, E_SNAPSHOT = range(13)
_E_ROWS = {E_EPISODE: 2, E_SNAPSHOT: 5}  # fields laid out [rows][n_env]
_E_INT = {E_DONE, E_FAIL, E_END, E_CUR_T, E_CONSUMED}
FRAME_STRIDE = 584
FR = dict(qpos=(0, 76), qvel=(76, 75), wbpos=(151, 72), wbquat=(223, 96), bquat=(319, 96), bangvel=(415, 72), ee_wpos=(487, 15), com=(502, 3), body_com=(512, 72))


def pack_expert_frames(feat) -> np.ndarray:
    """Expert feature dict (Humanoid.qpos_fk output) -> (T, FRAME_STRIDE) frame records of the clip bank."""
    T = int(feat["len"])
    out = np.zeros((T, FRAME_STRIDE))
    for k, (o, n) in FR.items():
        out[:, o:o + n] = np.asarray(feat[k]).reshape(T, n)
    return out


def expert_pose_of_frames(frames, ball: bool):
    """(qpos, qvel) of the model's own coordinates from frame records of the clip bank (numpy or torch, rows = frames): hinge angles as
    they are; ball joints -- root pose + the joints' quaternions (the record's body quaternions behind the root's), as the device-side
    reset builds it (uhc_env.hip: expert_qpos)."""
    import torch as _t
    cat = _t.cat if isinstance(frames, _t.Tensor) else np.concatenate
    q0, qn = FR["qpos"]
    b0, bn = FR["bquat"]
    v0, vn = FR["qvel"]
    qpos = cat([frames[:, q0:q0 + 7], frames[:, b0 + 4:b0 + bn]], 1) if ball else frames[:, q0:q0 + qn]
    return qpos, frames[:, v0:v0 + vn]


class EnvBatch:
    """Device env layer (uhc_env_* of the C-ABI) on top of a SimBatch."""

    def __init__(self, sim: SimBatch, desc: UhcEnvDesc):
        self.sim, self.L, self.desc = sim, sim.L, desc
        self._e = C.c_void_p()
        check(self.L.uhc_env_create(sim._b, C.byref(desc), C.byref(self._e)))
        self.obs_dim = self.L.uhc_env_obs_dim(self._e)
        self.n_env, self.device = sim.n_env, sim.device
        self._fields = {}
        self._bank = None

    def close(self):
        if getattr(self, "_e", None) is not None and self._e:
            if _capturing():
                _DEFERRED.append((self.L, "env", self._e, []))
                self._e = None
                return
            self.L.uhc_env_free(self._e)
            self._e = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def field(self, f: int) -> torch.Tensor:
        if f in self._fields:
            return self._fields[f]
        p, n = C.c_void_p(), C.c_int64()
        check(self.L.uhc_env_field(self._e, f, C.byref(p), C.byref(n)))
        per = n.value // self.n_env
        if f in _E_INT:
            t = torch.as_tensor(_DevView(p.value, (self.n_env,), "<i4", self), device=self.device)
        elif f in _E_ROWS:
            t = torch.as_tensor(_DevView(p.value, (_E_ROWS[f], self.n_env), "<f8", self), device=self.device)
        elif per == 1:
            t = torch.as_tensor(_DevView(p.value, (self.n_env,), "<f8", self), device=self.device)
        else:
            t = torch.as_tensor(_DevView(p.value, (self.n_env, per), "<f8", self), device=self.device)
        self._fields[f] = t
        return t

    generation = 0  # bumped whenever a device pointer the env kernels take BY VALUE changes (clip bank, clip -> model map): HIP graphs that
                    # captured env launches hold the old pointers and must be captured again (khrylib.rl.agents.Agent.rollout_step)

    def set_bank(self, frames: torch.Tensor, clip_start: torch.Tensor, clip_beta: torch.Tensor):
        self.generation += 1
        frames = frames.to(self.device, torch.float64).contiguous()
        clip_start = clip_start.to(self.device, torch.int32).contiguous()
        clip_beta = clip_beta.to(self.device, torch.float64).contiguous()
        assert frames.shape[1] == FRAME_STRIDE and clip_beta.shape == (clip_start.shape[0], 17)
        check(self.L.uhc_env_set_bank(self._e, C.c_void_p(frames.data_ptr()), frames.shape[0], C.c_void_p(clip_start.data_ptr()),
                                      C.c_void_p(clip_beta.data_ptr()), clip_start.shape[0]))
        self._bank = (frames, clip_start, clip_beta)  # keep alive: the library borrows the pointers

    def set_obj_pose(self, obj_pose: Optional[torch.Tensor]):
        """expert["obj_pose"] of every frame of the bank ([n_frames][7 num_obj]): where a reset puts the objects (humanoid_im.py:1284-1287)."""
        self.generation += 1
        if obj_pose is None:
            check(self.L.uhc_env_set_obj_pose(self._e, None, 0))
            self._obj_pose = None
            return
        op = obj_pose.to(self.device, torch.float64).contiguous()
        assert op.ndim == 2 and op.shape[0] == self._bank[0].shape[0] and op.shape[1] == 7 * int(self.desc.num_obj)
        check(self.L.uhc_env_set_obj_pose(self._e, C.c_void_p(op.data_ptr()), op.shape[0]))
        self._obj_pose = op  # borrowed by the library

    def set_clip_models(self, clip_model: Optional[torch.Tensor]):
        """Which of the batch's models the episodes of each clip run on (per-clip body shape); None switches it off."""
        self.generation += 1
        if clip_model is None:
            check(self.L.uhc_env_set_clip_models(self._e, None))
            self._clip_model = None
            return
        cm = clip_model.to(self.device, torch.int32).contiguous()
        assert cm.shape == (self._bank[1].shape[0],)
        check(self.L.uhc_env_set_clip_models(self._e, C.c_void_p(cm.data_ptr())))
        self._clip_model = cm  # borrowed by the library

    @staticmethod
    def _i32(t, device):
        return t.to(device, torch.int32).contiguous()

    def assign(self, env_ids, clip_ids, fr_start, fr_len):
        a, b, c, d = (self._i32(x, self.device) for x in (env_ids, clip_ids, fr_start, fr_len))
        check(self.L.uhc_env_assign(self._e, C.c_void_p(a.data_ptr()), a.shape[0], C.c_void_p(b.data_ptr()), C.c_void_p(c.data_ptr()),
                                    C.c_void_p(d.data_ptr())))

    def reset(self, env_ids, noise: Optional[torch.Tensor] = None):
        ids = self._i32(env_ids, self.device)
        nz = None
        if noise is not None:
            noise = noise.to(self.device, torch.float64).contiguous()
            assert noise.shape == (ids.shape[0], self.sim.model.nu)
            nz = C.c_void_p(noise.data_ptr())
        check(self.L.uhc_env_reset(self._e, C.c_void_p(ids.data_ptr()), ids.shape[0], nz))
        self._keep = (ids, noise)

    def set_next(self, env_ids, clip_ids, fr_start, fr_len, noise: Optional[torch.Tensor] = None):
        """Queue the window (and reset noise) the listed envs start when their current episode is done (uhc_env_set_next)."""
        a, b, c, d = (self._i32(x, self.device) for x in (env_ids, clip_ids, fr_start, fr_len))
        nz = None
        if noise is not None:
            noise = noise.to(self.device, torch.float64).contiguous()
            assert noise.shape == (a.shape[0], self.sim.model.nu)
            nz = C.c_void_p(noise.data_ptr())
        check(self.L.uhc_env_set_next(self._e, C.c_void_p(a.data_ptr()), a.shape[0], C.c_void_p(b.data_ptr()), C.c_void_p(c.data_ptr()),
                                      C.c_void_p(d.data_ptr()), nz))
        self._keep_next = (a, b, c, d, noise)  # alive until the stream has consumed them (the next call replaces them a step later)

    def set_next_host(self, env_ids, clip_ids, fr_start, fr_len, noise=None):
        """set_next from host (numpy) arrays: the five arrays are packed into one buffer and go down in ONE copy (a copy from pageable
        memory waits for everything queued before it; five of them per control step cost ~0.1 ms of idle GPU).  Measured alternatives,
        both dropped: an asynchronous copy from a pinned ring, and the kernel reading the pinned buffer over PCIe itself -- with either
        the host runs ahead of the GPU and the step time became erratic (4.1 .. 8 ms against a steady 4.1)."""
        import numpy as np
        k, nu = int(len(env_ids)), self.sim.model.nu
        if k == 0:
            return
        nbytes = 16 * k + (8 * k * nu if noise is not None else 0)
        hb = np.empty(nbytes, dtype=np.uint8)
        ints = hb[:16 * k].view(np.int32).reshape(4, k)
        ints[0], ints[1], ints[2], ints[3] = env_ids, clip_ids, fr_start, fr_len
        if noise is not None:
            hb[16 * k:].view(np.float64).reshape(k, nu)[...] = noise
        dev = self.__dict__.get("_next_dev")
        if dev is None or dev.numel() < nbytes:
            dev = self._next_dev = torch.empty(max(nbytes, 16 * self.n_env + 8 * self.n_env * nu), dtype=torch.uint8, device=self.device)
        dev[:nbytes].copy_(torch.from_numpy(hb))
        base = dev.data_ptr()
        check(self.L.uhc_env_set_next(self._e, C.c_void_p(base), k, C.c_void_p(base + 4 * k), C.c_void_p(base + 8 * k), C.c_void_p(base + 12 * k),
                                      C.c_void_p(base + 16 * k) if noise is not None else None))

    def auto_reset(self):
        """Device-side episode turnover of every env whose done flag is set (uhc_env_auto_reset); no host round trip."""
        check(self.L.uhc_env_auto_reset(self._e))

    def set_end_reward(self, v: float):
        check(self.L.uhc_env_set_end_reward(self._e, float(v)))

    def step(self, action: torch.Tensor, active: Optional[torch.Tensor] = None):
        assert action.dtype == torch.float64 and action.is_contiguous() and action.shape == (self.n_env, self.sim.ctrl.action_dim)
        a = None
        if active is not None:
            assert active.dtype == torch.int32 and active.is_contiguous()
            a = C.c_void_p(active.data_ptr())
        check(self.L.uhc_env_step(self._e, C.c_void_p(action.data_ptr()), a))


def make_ctrl(model, *, meta_pd: bool = True, meta_pd_joint: bool = False, residual_force: bool = True,
              residual_force_mode: str = "implicit", residual_force_scale: float = 100.0, residual_force_lim: float = 100.0,
              rfc_rate: float = 1.0, action_type: str = "position", pd_mul: float = 1.0, tq_mul: float = 1.0,
              base_rot=(0.7071, 0.7071, 0.0, 0.0), n_substeps: int = 15, residual_force_bodies="all",
              residual_force_torque: bool = True) -> UhcCtrlDesc:
    """UhcCtrlDesc from the reference's config knobs (copycat_config.py:86-113, humanoid_im.py:120-124,226-255)."""
    from ._capi import ctrl_desc
    from .smpllib.smpl_mujoco import SMPLConverter

    conv = SMPLConverter(model, model)
    nu = model.nu
    rfc_mode = 0
    vf_dim = 0
    vf_body, body_vf_dim, scale = None, 9, residual_force_scale * rfc_rate
    if residual_force:
        if residual_force_mode == "implicit":
            rfc_mode, vf_dim = 1, 6
        else:  # explicit: one (contact point, force[, torque]) per body, bodies in SMPL joint order (humanoid_im.py:236-243)
            from .smpllib.smpl_mujoco import SMPL_BONE_ORDER_NAMES
            names = SMPL_BONE_ORDER_NAMES if residual_force_bodies == "all" else list(residual_force_bodies)
            vf_body = [model.body_names.index(n) for n in names]
            body_vf_dim = 6 + 3 * int(bool(residual_force_torque))
            rfc_mode, vf_dim, scale = 2, body_vf_dim * len(vf_body), residual_force_scale  # rfc_explicit does not apply rfc_rate
    mp = 1 if meta_pd else (2 if meta_pd_joint else 0)
    mp_dim = 2 * n_substeps if mp == 1 else (2 * nu if mp == 2 else 0)
    return ctrl_desc(n_substeps=n_substeps, action_type=0 if action_type == "position" else 1, meta_pd=mp, rfc_mode=rfc_mode,
                     action_dim=nu + vf_dim + mp_dim, rfc_scale=scale, rfc_lim=residual_force_lim, vf_body=vf_body, body_vf_dim=body_vf_dim,
                     base_rot=base_rot, jkp=conv.get_new_jkp() * pd_mul, jkd=conv.get_new_jkd() * pd_mul,
                     torque_lim=conv.get_new_torque_limit() * tq_mul, a_scale=conv.get_new_a_scale())


def load_asset_model(name: str = "humanoid_smpl_neutral_mesh"):
    import os
    from .model.mjcf import Model
    return Model.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", name + ".npz"))
