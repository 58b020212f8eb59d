"""GPU diagnostic: the general kernel's working-set solve vs the oracle's exact solve on forward passes and a short trajectory."""
import dataclasses, os, sys
os.environ["UHC_FORCE_GENERAL"] = "1"
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.physics import OracleSim
from uhc_amd import sim as S
model = dataclasses.replace(S.load_asset_model(), solver=1)
ctrl = S.make_ctrl(model)
z = dict(np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(21)
qpos = np.tile(z["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(scale=0.1, size=(n, 69)); qvel = rng.normal(scale=0.5, size=(n, 75))
b = S.SimBatch(model, ctrl, n)
b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel)); b.sync()
os_ = []
for e in range(n):
    o = OracleSim(model, ctrl); o.set_state(qpos[e], qvel[e]); os_.append(o)
    d = np.abs(b.field(S.F_QACC)[e].cpu().numpy() - o.get("qacc")).max()
    print("fwd env", e, "nefc", int(b.field(S.F_NEFC)[e]), o.geti("nefc"), "iters", int(b.field(S.F_SOLVER_ITER)[e]), o.geti("solver_iter"), "redo", int(b.field(S.F_REDO)[e]), "qacc diff %.3e" % d)
tb = torch.from_numpy(qpos[:, 7:].copy()).cuda()
for t in range(6):
    act = rng.normal(scale=0.05, size=(n, ctrl.action_dim))
    b.simulate(torch.from_numpy(act).cuda(), tb); b.sync()
    gq = b.field(S.F_QPOS).cpu().numpy(); redo = b.field(S.F_REDO).cpu().numpy()
    line = []
    for e in range(n):
        os_[e].do_simulation(act[e], qpos[e, 7:], redo=redo[e])
        line.append("%.1e/%d" % (np.abs(gq[e] - os_[e].get("qpos")).max(), redo[e]))
    print("step", t, " ".join(line))
