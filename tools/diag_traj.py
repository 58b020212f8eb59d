"""GPU diagnostic: the 200-step contact trajectory of tests/test_gpu_physics.py with solver 1, step by step: where the GPU and the oracle
part, and what UHC_F_REDO says about those steps."""
import dataclasses, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.physics import OracleSim
from uhc_amd import sim as S
model = dataclasses.replace(S.load_asset_model(), solver=1)
ctrl = S.make_ctrl(model)
z = dict(np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz")))
n = 4
rng = np.random.default_rng(4)
qpos = np.tile(z["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(scale=0.02, size=(n, model.nu)); qvel = rng.normal(scale=0.05, size=(n, model.nv))
rng = np.random.default_rng(5)
b = S.SimBatch(model, ctrl, n)
b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
os_ = [OracleSim(model, ctrl) for _ in range(n)]
for e in range(n):
    os_[e].set_state(qpos[e], qvel[e])
for t in range(200):
    act = rng.normal(scale=0.05, size=(n, ctrl.action_dim))
    b.simulate(torch.from_numpy(act).cuda(), tb); b.sync()
    gq = b.field(S.F_QPOS).cpu().numpy(); redo = b.field(S.F_REDO).cpu().numpy(); nefc = b.field(S.F_NEFC).cpu().numpy(); it = b.field(S.F_SOLVER_ITER).cpu().numpy()
    line = []
    for e in range(n):
        os_[e].do_simulation(act[e], qpos[e, 7:], redo=redo[e])
        d = np.abs(gq[e] - os_[e].get("qpos")).max()
        line.append("%.1e r%d n%d/%d i%d/%d" % (d, redo[e], nefc[e], os_[e].geti("nefc"), it[e], os_[e].geti("solver_iter")))
    if redo.any() or t % 20 == 0:
        print("step", t, " | ".join(line))
