"""Diagnostic for tests/test_gpu_selfcollision.py::test_a_dof_chain_of_32_entries_goes_through_tier_4: per step flags, rows and error against the oracle."""
import dataclasses, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import caterpillar_model, passive_ctrl
from oracle.physics import OracleSim
from uhc_amd import sim as S

m = dataclasses.replace(caterpillar_model(27), solver=1)
ctrl = passive_ctrl(m, n_substeps=15)
rng = np.random.default_rng(3)
def pose(lifted):
    q = m.qpos0.copy(); q[2] -= 0.0004
    q[7:] = rng.normal(scale=1e-4, size=m.nq - 7)
    if lifted: q[7 + 5] = -1.1
    return q
for vs in (0.05, 0.0):
    n = 2
    q = np.stack([pose(False) for _ in range(n)])
    v = rng.normal(scale=vs, size=(n, m.nv))
    b = S.SimBatch(m, ctrl, n)
    b.set_state(torch.from_numpy(q), torch.from_numpy(v)); b.sync()
    os_ = [OracleSim(m, ctrl) for _ in range(n)]
    for e in range(n):
        os_[e].set_state(q[e], v[e])
        print("vs", vs, "env", e, "nefc dev/orc", int(b.field(S.F_NEFC)[e].item()), os_[e].geti("nefc"), "qacc err", np.abs(b.field(S.F_QACC)[e].cpu().numpy() - os_[e].get("qacc")).max())
    act = np.zeros((n, ctrl.action_dim)); tb = torch.zeros(n, max(m.nu, 1), dtype=torch.float64, device="cuda")
    for t in range(4):
        b.simulate(torch.from_numpy(act).cuda(), tb); b.sync()
        redo = b.field(S.F_REDO).cpu().numpy(); why = b.field(S.F_HANDON_WHY).cpu().numpy()
        for e in range(n):
            os_[e].do_simulation(act[e], np.zeros(max(m.nu, 1)))
            print("  t", t, "env", e, "redo", hex(int(redo[e])), "why", hex(int(why[e])), "nefc", int(b.field(S.F_NEFC)[e].item()), os_[e].geti("nefc"), "fail", int(b.field(S.F_FAIL)[e].item()),
                  "err", np.abs(b.field(S.F_QPOS)[e].cpu().numpy() - os_[e].get("qpos")).max(), "z", os_[e].get("qpos")[2])
    b.close()
