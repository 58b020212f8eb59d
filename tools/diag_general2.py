"""GPU diagnostic: general-kernel forward after a fast batch lived in the same process (as the test suite does)."""
import dataclasses, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.physics import OracleSim
from uhc_amd import sim as S
model = dataclasses.replace(S.load_asset_model(), solver=1)
ctrl = S.make_ctrl(model)
z = dict(np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz")))
n = 32
rng = np.random.default_rng(21)
qpos = np.tile(z["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(scale=0.1, size=(n, 69)); qvel = rng.normal(scale=0.5, size=(n, 75))
for path in ("0", "1", "1"):
    os.environ["UHC_FORCE_GENERAL"] = path
    b = S.SimBatch(model, ctrl, n)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel)); b.sync()
    bad = []
    for e in range(n):
        o = OracleSim(model, ctrl); o.set_state(qpos[e], qvel[e])
        d = np.abs(b.field(S.F_QACC)[e].cpu().numpy() - o.get("qacc")).max()
        if d > 1e-7:
            bad.append((e, int(b.field(S.F_NEFC)[e]), o.geti("nefc"), int(b.field(S.F_SOLVER_ITER)[e]), o.geti("solver_iter"), int(b.field(S.F_REDO)[e]), int(b.field(S.F_EFC_OVERFLOW)[e]), float(d)))
    print("path", path, "bad envs:", bad)
    b.close()
