"""Diagnostic for tests/test_gpu_rounded_hulls.py: error of every env per control step, device against oracle (free-running and re-started from the device's state)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_rounded_hulls import foot_model, _poses
from tests.helpers import passive_ctrl
from oracle.physics import OracleSim
from uhc_amd import sim as S

m = foot_model()
n = 16
q = _poses(m, n, 11)
v = np.random.default_rng(12).normal(scale=0.3, size=(n, m.nv))
ctrl = passive_ctrl(m, n_substeps=5)
b = S.SimBatch(m, ctrl, n)
b.set_state(torch.from_numpy(q), torch.from_numpy(v))
free = [OracleSim(m, ctrl) for _ in range(n)]
sync = [OracleSim(m, ctrl) for _ in range(n)]
for e in range(n):
    free[e].set_state(q[e], v[e]); sync[e].set_state(q[e], v[e])
act = torch.zeros(n, ctrl.action_dim, dtype=torch.float64, device="cuda")
tb = torch.zeros(n, max(m.nu, 1), dtype=torch.float64, device="cuda")
z = np.zeros(ctrl.action_dim); zt = np.zeros(max(m.nu, 1))
for t in range(40):
    b.simulate(act, tb); b.sync()
    gq = b.field(S.F_QPOS).cpu().numpy(); gv = b.field(S.F_QVEL).cpu().numpy(); redo = b.field(S.F_REDO).cpu().numpy()
    ef, es = np.zeros(n), np.zeros(n)
    for e in range(n):
        for o in (free[e], sync[e]):
            o.desc.solver = 0 if (int(redo[e]) & 2) else 1
            o.do_simulation(z, zt)
        ef[e] = np.abs(gq[e] - free[e].get("qpos")).max(); es[e] = np.abs(gq[e] - sync[e].get("qpos")).max()
        nefc = sync[e].geti("nefc")
        sync[e].set_state(gq[e], gv[e])
    k = int(np.argmax(es))
    print(f"t {t:2d} free-running worst {ef.max():.2e} (env {int(np.argmax(ef))}) | re-started each step worst {es.max():.2e} (env {k}, redo {hex(int(redo[k]))}, nefc {int(b.field(S.F_NEFC)[k].item())})")
b.close()
