import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle.physics import OracleSim
from uhc_amd import sim as S
model = S.load_asset_model()
z = np.load("uhc_amd/assets/standing_neutral.npz")
lift = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
for nsub in (1, 2, 15):
    ctrl = S.make_ctrl(model, n_substeps=nsub)
    # keep action layout of 15 substeps irrelevant: meta_pd dims = 2*nsub
    rng = np.random.default_rng(2)
    n = 2
    qpos = np.tile(z["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(scale=0.1, size=(n, 69)); qpos[:, 2] += lift
    qvel = rng.normal(scale=0.5, size=(n, 75))
    act = rng.normal(scale=0.2, size=(n, ctrl.action_dim))
    b = S.SimBatch(model, ctrl, n)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    a, tb = torch.from_numpy(act).cuda(), torch.from_numpy(qpos[:, 7:].copy()).cuda()
    o = OracleSim(model, ctrl); o.set_state(qpos[0], qvel[0])
    for rep in range(3):
        b.simulate(a, tb); b.sync()
        o.do_simulation(act[0], qpos[0, 7:])
        print(f"nsub={nsub} rep={rep}", end=" ")
        for name, f in (("ctrl", S.F_CTRL), ("applied", S.F_QFRC_APPLIED), ("qacc", S.F_QACC), ("qvel", S.F_QVEL), ("qpos", S.F_QPOS), ("bias", S.F_QFRC_BIAS), ("qM", S.F_QM)):
            g = b.field(f)[0].cpu().numpy()
            r = o.get({"ctrl": "ctrl", "applied": "qfrc_applied", "qacc": "qacc", "qvel": "qvel", "qpos": "qpos", "bias": "qfrc_bias", "qM": "qM"}[name])
            print(f"{name}:{np.abs(g - r).max():.2e}", end=" ")
        print("ncon", b.field(S.F_NCON)[0].item(), o.geti("ncon"), "nefc", b.field(S.F_NEFC)[0].item(), o.geti("nefc"), "it", b.field(S.F_SOLVER_ITER)[0].item(), o.geti("solver_iter"))
    b.close()
