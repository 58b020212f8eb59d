import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle.physics import OracleSim
from uhc_amd import sim as S
model = S.load_asset_model()
z = np.load("uhc_amd/assets/standing_neutral.npz")
lift = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
ctrl = S.make_ctrl(model)
rng = np.random.default_rng(2)
n = 4
qpos = np.tile(z["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(scale=0.1, size=(n, 69)); qpos[:, 2] += lift
qvel = rng.normal(scale=0.5, size=(n, 75))
rng = np.random.default_rng(3)
act = rng.normal(scale=0.2, size=(n, ctrl.action_dim))
b = S.SimBatch(model, ctrl, n)
b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
a, tb = torch.from_numpy(act).cuda(), torch.from_numpy(qpos[:, 7:].copy()).cuda()
os_ = [OracleSim(model, ctrl) for _ in range(n)]
for e in range(n): os_[e].set_state(qpos[e], qvel[e])
for t in range(50):
    b.simulate(a, tb); b.sync()
    gq = b.field(S.F_QPOS).cpu().numpy(); gv = b.field(S.F_QVEL).cpu().numpy(); gc = b.field(S.F_CTRL).cpu().numpy()
    line = f"t={t} "
    for e in range(n):
        os_[e].do_simulation(act[e], qpos[e, 7:])
        line += f"[{np.abs(gq[e]-os_[e].get('qpos')).max():.1e} {np.abs(gv[e]-os_[e].get('qvel')).max():.1e} sat={int((np.abs(gc[e])>=ctrl_lim).sum()) if (ctrl_lim:=np.array([ctrl.torque_lim[i] for i in range(69)])) is not None else 0} f={b.field(S.F_FAIL)[e].item()}] "
    print(line)
