"""Two exact solvers, one optimum: how far apart do they land in ONE control step?  The oracle's dual active-set solve (solver 1) against its primal Newton solve
(solver 2) -- both stop at tolerances (KKT 1e-9 (1 + max |b|) / a Newton step that leaves the active set alone or |g| <= 1e-14 |g0|) -- on the 200-step parity
workloads (tests/test_gpu_parity_200.py): the trajectory runs on solver 1; at every control step a twin is started from the same state with solver 2 and the two
results are compared.  This is the per-step injection that two IMPLEMENTATIONS of the exact solve (device / oracle) cannot be expected to stay under -- DESIGN 2 "why
one control step of ball_objects differs by 1e-10, not 1e-14" reasoned it from the stopping rules; this measures it.   python tools/solver_spread.py > profiles/r06_solver_spread.txt"""
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.physics import OracleSim  # noqa: E402
from tests.test_gpu_parity_200 import _class  # noqa: E402
from uhc_amd.sim import load_asset_model  # noqa: E402

model = load_asset_model()
z = np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz"))
standing = {k: z[k] for k in z.files}
print(__doc__)
STEPS = int(os.environ.get("STEPS", "120"))
# (the torque-driven classes only: a twin started with set_state carries a FRESH mass matrix and bias, while the PD controller of the hinge classes reads the ones the last
#  substep left behind -- humanoid_im.py:1019-1022's stale-M semantics --, which moves a control step by 5e-6 whatever the solver)
for name in ("ball", "ball_objects"):
    m, ctrl, q0, v0, a_sc, tb = _class(name, model, standing)
    m2 = dataclasses.replace(m, solver=2)
    n = q0.shape[0]
    A = [OracleSim(m, ctrl) for _ in range(n)]
    for e in range(n):
        A[e].set_state(q0[e], v0[e])
    rng = np.random.default_rng(7)
    dq, dv, rows = np.zeros((STEPS, n)), np.zeros((STEPS, n)), np.zeros((STEPS, n), dtype=int)
    for t in range(STEPS):
        act = rng.normal(scale=a_sc, size=(n, ctrl.action_dim))
        for e in range(n):
            q, v = A[e].get("qpos").copy(), A[e].get("qvel").copy()
            B = OracleSim(m2, ctrl)
            B.set_state(q, v)
            B.set("qacc_warmstart", A[e].get("qacc_warmstart"))
            A[e].do_simulation(act[e], tb[e])
            B.do_simulation(act[e], tb[e])
            dq[t, e] = np.abs(A[e].get("qpos") - B.get("qpos")).max()
            dv[t, e] = np.abs(A[e].get("qvel") - B.get("qvel")).max()
            rows[t, e] = A[e].geti("nefc")
    pct = lambda a: " ".join(f"{x:.1e}" for x in np.percentile(a, [50, 90, 99, 100]))
    print(f"{name:13s} {STEPS} steps x {n} envs, rows at the steps' ends {rows.min()}..{rows.max()}: one control step, dual active set vs primal Newton from the same state -- "
          f"|dqpos| median / 90 % / 99 % / max {pct(dq)};  |dqvel| {pct(dv)};  steps beyond 1e-9 in qvel: {int((dv > 1e-9).sum())} of {dv.size} "
          f"(a difference of 1e-12 in one substep picks another vertex in a later one: the ties of DESIGN 2)")
