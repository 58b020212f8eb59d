import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle.physics import OracleSim
from uhc_amd import sim as S
model = S.load_asset_model()
z = np.load("uhc_amd/assets/standing_neutral.npz")
ctrl = S.make_ctrl(model, n_substeps=1)
rng = np.random.default_rng(0)
n = 2
qpos = np.tile(z["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(scale=0.05, size=(n, 69))
qvel = rng.normal(scale=0.1, size=(n, 75))
act = rng.normal(scale=0.1, size=(n, ctrl.action_dim))
b = S.SimBatch(model, ctrl, n)
b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
a, tb = torch.from_numpy(act).cuda(), torch.from_numpy(qpos[:, 7:].copy()).cuda()
o = OracleSim(model, ctrl); o.set_state(qpos[0], qvel[0])
print("after set_state qacc diff", np.abs(b.field(S.F_QACC)[0].cpu().numpy() - o.get("qacc")).max(), "nefc", b.field(S.F_NEFC)[0].item(), o.geti("nefc"), "it", b.field(S.F_SOLVER_ITER)[0].item(), o.geti("solver_iter"))
for t in range(60):
    b.simulate(a, tb); b.sync()
    o.do_simulation(act[0], qpos[0, 7:])
    d = {k: np.abs(b.field(f)[0].cpu().numpy() - o.get(nm)).max() for k, f, nm in (("qacc", S.F_QACC, "qacc"), ("qvel", S.F_QVEL, "qvel"), ("qpos", S.F_QPOS, "qpos"), ("ctrl", S.F_CTRL, "ctrl"))}
    print(t, " ".join(f"{k}:{v:.1e}" for k, v in d.items()), "ncon", b.field(S.F_NCON)[0].item(), o.geti("ncon"), "nefc", b.field(S.F_NEFC)[0].item(), o.geti("nefc"), "it", b.field(S.F_SOLVER_ITER)[0].item(), o.geti("solver_iter"))
