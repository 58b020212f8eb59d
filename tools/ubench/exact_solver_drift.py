"""Diagnostic: per-step GPU-vs-oracle difference with solver 1 (exact active-set solve), to separate rounding amplification
(smooth exponential growth) from a discrete event (a jump).  Run on the GPU box."""
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle.build import build  # noqa: E402

build()
from oracle.physics import OracleSim  # noqa: E402
from uhc_amd import sim as S  # noqa: E402

solver = int(os.environ.get("SOLVER", "1"))
model = dataclasses.replace(S.load_asset_model(), solver=solver, iterations=int(os.environ.get("CAP", "100")))
ctrl = S.make_ctrl(model)
z = np.load(os.path.join(os.path.dirname(S.__file__), "assets", "standing_neutral.npz"))
n = 4
rng = np.random.default_rng(4)
qpos = np.tile(z["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(scale=0.02, size=(n, model.nu))
qvel = rng.normal(scale=0.05, size=(n, model.nv))
rng = np.random.default_rng(5)
b = S.SimBatch(model, ctrl, n)
b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
os_ = [OracleSim(model, ctrl) for _ in range(n)]
for e in range(n):
    os_[e].set_state(qpos[e], qvel[e])
for t in range(200):
    act = rng.normal(scale=0.05, size=(n, ctrl.action_dim))
    b.simulate(torch.from_numpy(act).cuda(), tb)
    b.sync()
    gq = b.field(S.F_QPOS).cpu().numpy()
    git, gn = b.field(S.F_SOLVER_ITER).cpu().numpy(), b.field(S.F_NEFC).cpu().numpy()
    d = []
    for e in range(n):
        os_[e].do_simulation(act[e], qpos[e, 7:])
        d.append(np.abs(gq[e] - os_[e].get("qpos")).max())
    if t % 10 == 0 or t > 190:
        print(t, " ".join(f"{x:.1e}" for x in d), "iters gpu", git.tolist(), "oracle", [o.geti("solver_iter") for o in os_], "nefc", gn.tolist(), flush=True)
