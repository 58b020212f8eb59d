"""Tier 4 alone: a batch of humanoids face down beside the seven-box raft (tests/test_gpu_selfcollision.py: 300-330 rows, 150+ with a force), every env
of it beyond the large tier from the first forward pass on -- per-control-step wall time (HIP events are the fast tier's; here the whole step is
synchronised) and, on the instrumented library (UHC_LIB=uhc_amd/csrc/libuhc_amd_prof.so), the stage cycles of the Newton iteration.
    python tools/bench_tier4.py [n_env] [steps]
KPATH=2: sticky tiers -- after the first step every env starts in the large tier, whose consumers hand it to tier 4's FOUR-WAVE consumers (uhc_k_huge_q.hip);
give them a workgroup per env (UHC_Q4_MAX=128) to time the four-wave Newton iteration itself.  Default: the tier chain (tier 4 on one wave, behind the large
tier's workgroup)."""
import dataclasses
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from uhc_amd import sim as S  # noqa: E402
from uhc_amd.model.mjcf import add_free_bodies, quat_mul, self_collision_variant  # noqa: E402
from uhc_amd.model.shapes import box_triangles  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
# EXTRA=k (with KPATH=2): k more envs STANDING beside the raft -- with UHC_TIER_MARKS="16,4,2,8,2,1,8,7" they live in the general tier, so that the host starts the
# queue consumers of every tier, and with UHC_T4_ROWS=200 UHC_Q4_MAX=<n> the face-down envs start every step in tier 4's four-wave consumers
extra = int(os.environ.get("EXTRA", "0"))
n_down = n
n += extra
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
model = S.load_asset_model()
z = np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz"))
K = 7
m = self_collision_variant(model)
yaw = [0.06 * (-1) ** k for k in range(K)]
poses = np.array([[1.0 + 0.305 * k, 1.0 + 0.01 * k, 0.1495, np.cos(y / 2), 0, 0, np.sin(y / 2)] for k, y in enumerate(yaw)], dtype=np.float64)
m = dataclasses.replace(add_free_bodies(m, [box_triangles(0.15, 0.15, 0.15)] * K, poses, density=5.0 / 0.027), solver=1)
ctrl = S.make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
rng = np.random.default_rng(0)
q = np.tile(m.qpos0, (n, 1))
for e in range(n):
    qh = z["qpos"].copy()
    qh[7:] += rng.normal(scale=0.02, size=69)
    if e < n_down:
        a = np.pi / 2 + 0.1 * rng.uniform(-1, 1)
        qh[3:7] = quat_mul(np.array([np.cos(a / 2), 0, np.sin(a / 2), 0]), qh[3:7])
        qh[0], qh[1], qh[2] = -0.8, -0.8, 0.14
    else:
        qh[0], qh[1] = -2.0, -2.0
    q[e, :76] = qh
b = S.SimBatch(m, ctrl, n)
if os.environ.get("KPATH"):
    b.set_kernel_path(int(os.environ["KPATH"]))
b.set_state(torch.from_numpy(q), torch.zeros(n, m.nv, dtype=torch.float64))
b.sync()
print(f"{n} envs, nv {m.nv}; after set_state: nefc mean {b.field(S.F_NEFC).float().mean().item():.0f} max {int(b.field(S.F_NEFC).max().item())}, Newton iterations of that pass mean "
      f"{b.field(S.F_SOLVER_ITER).float().mean().item():.1f} max {int(b.field(S.F_SOLVER_ITER).max().item())}")
prof = os.environ.get("UHC_LIB", "").endswith("_prof.so")
if prof:
    b.field(S.F_STAGE_PROF).zero_()
act = torch.zeros(n, ctrl.action_dim, dtype=torch.float64, device="cuda")
tb = torch.zeros(n, 69, dtype=torch.float64, device="cuda")
for t in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.simulate(act, tb)
    b.sync()
    ms = 1e3 * (time.perf_counter() - t0)
    redo = b.field(S.F_REDO).cpu().numpy()
    print(f"step {t}: {ms:7.2f} ms; through tier 4: {int(((redo & (1 << 30)) != 0).sum())} envs, cap hits {int(((redo & (1 << 29)) != 0).sum())}, rows dropped {int(((redo & 0x80) != 0).sum())}; "
          f"nefc (last substep) mean {b.field(S.F_NEFC).float().mean().item():.0f} max {int(b.field(S.F_NEFC).max().item())}; Newton iterations (last substep) mean "
          f"{b.field(S.F_SOLVER_ITER).float().mean().item():.1f} max {int(b.field(S.F_SOLVER_ITER).max().item())}")
if prof:
    p = b.field(S.F_STAGE_PROF).cpu().numpy().astype(np.float64)[:n_down] / steps
    names = {0: "pd+rfc / torque", 1: "kinematics", 2: "com_pos", 3: "crb", 4: "factor", 5: "com_vel", 6: "rne", 7: "smooth", 8: "collision", 9: "rows (to HBM)",
             11: "solve: rest", 13: "qacc", 14: "euler", 15: "store", 30: "newton: start point (u0, jar0, cost)", 31: "newton: jar, gradient, Hessian build", 26: "newton: Cholesky",
             27: "newton: substitutions", 28: "newton: p = Yhat dir, line search", 32: "col: plane-mesh", 33: "col: cull", 34: "col: staging", 35: "col: MPR",
             36: "newton: jar = Yhat u + b", 37: "newton: active set + rank-one updates", 38: "newton: chain rows -> gradient (+ Hessian)", 39: "newton: dense rows -> gradient + Hessian"}
    names[31] = "newton: gradient norm / rest"
    tot = p.sum(1)
    print(f"instrumented: mean cycles per env-step {tot.mean():.3e} (all tiers' work on the env, the large tier's abandoned passes included)")
    for k in sorted(names):
        if p[:, k].mean() > 0:
            print(f"  {names[k]:40s} {p[:, k].mean():12.0f} cycles/env-step {100 * p[:, k].mean() / tot.mean():5.1f}%")
