"""GPU diagnostic: ball-joint humanoid + objects vs the oracle, step by step (run through gpurun)."""
import dataclasses, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.physics import OracleSim
from uhc_amd import sim as S
from tests.test_gpu_ball import _ball_setup, _states
model = S.load_asset_model()
standing = dict(np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz")))
objects, selfc = int(sys.argv[1]) if len(sys.argv) > 1 else 3, (sys.argv[2] == "1") if len(sys.argv) > 2 else True
ball, ctrl = _ball_setup(model, objects=objects, self_collision=selfc)
n = 3
q, v = _states(model, ball, standing, n, 43, 0.0)
b = S.SimBatch(ball, ctrl, n)
b.set_state(torch.from_numpy(q), torch.from_numpy(v)); b.sync()
os_ = [OracleSim(ball, ctrl) for _ in range(n)]
redo = b.field(S.F_REDO).cpu().numpy()
for e in range(n):
    os_[e].desc.solver = 0 if redo[e] else 1
    os_[e].set_state(q[e], v[e])
    print("init env", e, "ncon", int(b.field(S.F_NCON)[e]), os_[e].geti("ncon"), "nefc", int(b.field(S.F_NEFC)[e]), os_[e].geti("nefc"), "redo", redo[e],
          "qacc diff", np.abs(b.field(S.F_QACC)[e].cpu().numpy() - os_[e].get("qacc")).max())
rng = np.random.default_rng(44)
tb = torch.zeros(n, 69, dtype=torch.float64, device="cuda")
for t in range(12):
    act = rng.normal(scale=0.003, size=(n, ctrl.action_dim))
    b.simulate(torch.from_numpy(act).cuda(), tb); b.sync()
    gq = b.field(S.F_QPOS).cpu().numpy(); redo = b.field(S.F_REDO).cpu().numpy()
    for e in range(n):
        os_[e].desc.solver = 0 if redo[e] else 1
        os_[e].do_simulation(act[e], np.zeros(69))
        d = np.abs(gq[e] - os_[e].get("qpos"))
        print(t, e, "diff %.3e at %d" % (d.max(), d.argmax()), "ncon", int(b.field(S.F_NCON)[e]), os_[e].geti("ncon"), "nefc", int(b.field(S.F_NEFC)[e]), os_[e].geti("nefc"),
              "max", os_[e].geti("max_ncon"), os_[e].geti("max_nefc"), "redo", redo[e], "ovf", int(b.field(S.F_EFC_OVERFLOW)[e]), "fail", int(b.field(S.F_FAIL)[e]), os_[e].geti("fail"))
