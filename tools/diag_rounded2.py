"""Find the first substep at which device and oracle part in tests/test_gpu_rounded_hulls.py's trajectory (env 3, control step 8) and print both sides' contacts."""
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
ctrl = passive_ctrl(m, n_substeps=1)
b = S.SimBatch(m, ctrl, n)
b.set_state(torch.from_numpy(q), torch.from_numpy(v)); b.sync()
act = torch.zeros(n, ctrl.action_dim, dtype=torch.float64, device="cuda")
tb = torch.zeros(n, max(m.nu, 1), dtype=torch.float64, device="cuda")
z = np.zeros(ctrl.action_dim); zt = np.zeros(max(m.nu, 1))
o = OracleSim(m, ctrl)
E = int(os.environ.get("ENV", "3"))
for t in range(60):
    pq, pv = b.field(S.F_QPOS).cpu().numpy().copy(), b.field(S.F_QVEL).cpu().numpy().copy()
    o.set_state(pq[E], pv[E])
    d_ncon, d_nefc = int(b.field(S.F_NCON)[E].item()), int(b.field(S.F_NEFC)[E].item())
    d_qacc = b.field(S.F_QACC)[E].cpu().numpy()
    b.simulate(act, tb); b.sync()
    o.do_simulation(z, zt)
    gq, gv = b.field(S.F_QPOS).cpu().numpy(), b.field(S.F_QVEL).cpu().numpy()
    err = max(np.abs(gq[E] - o.get("qpos")).max(), np.abs(gv[E] - o.get("qvel")).max())
    if err > 1e-8 or t % 10 == 0:
        print(f"substep {t}: err {err:.2e}; after the step: device ncon/nefc {int(b.field(S.F_NCON)[E].item())}/{int(b.field(S.F_NEFC)[E].item())} oracle {o.geti('ncon')}/{o.geti('nefc')} redo {hex(int(b.field(S.F_REDO)[E].item()))}")
    if err > 1e-8:
        o2 = OracleSim(m, ctrl); o2.set_state(pq[E], pv[E])
        print(" state before:", repr(pq[E]), repr(pv[E]))
        print(" oracle at the state before: ncon", o2.geti("ncon"), "nefc", o2.geti("nefc"))
        print("  geom1", o2.get("con_geom1"), "geom2", o2.get("con_geom2"))
        print("  dist", o2.get("con_dist"))
        print("  pos", o2.get("con_pos").reshape(-1, 3))
        print("  normal", o2.get("con_frame").reshape(-1, 9)[:, :3])
        print("  oracle qacc", o2.get("qacc"))
        b2 = S.SimBatch(m, ctrl, 1)
        b2.set_state(torch.from_numpy(pq[E:E + 1]), torch.from_numpy(pv[E:E + 1])); b2.sync()
        print("  device at the state before: ncon", int(b2.field(S.F_NCON)[0].item()), "nefc", int(b2.field(S.F_NEFC)[0].item()))
        print("  device qacc", b2.field(S.F_QACC)[0].cpu().numpy())
        print("  |dqacc|", np.abs(b2.field(S.F_QACC)[0].cpu().numpy() - o2.get("qacc")).max())
        break
