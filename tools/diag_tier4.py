"""Tier 4 substep by substep (n_substeps = 1): the face-down + raft scene, device (tier 4) against the oracle after every substep."""
import dataclasses, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.physics import OracleSim
from uhc_amd import sim as S
from tests.test_oracle_physics import _face_down_beside_a_raft
model = S.load_asset_model()
z = np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz")); standing = {k: z[k] for k in z.files}
m, ctrl15, q, v = _face_down_beside_a_raft(model, standing)
ctrl = S.make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4, n_substeps=1)
n = 2
b = S.SimBatch(m, ctrl, n)
b.set_state(torch.from_numpy(np.tile(q, (n, 1))), torch.from_numpy(np.tile(v, (n, 1))))
b.sync()
o = OracleSim(m, ctrl); o.set_state(q, v)
print("set_state: nefc", b.field(S.F_NEFC).tolist(), "iters", b.field(S.F_SOLVER_ITER).tolist(), "redo", [hex(int(x)) for x in b.field(S.F_REDO).tolist()], "oracle nefc", o.geti("nefc"), "iters", o.geti("solver_iter"),
      "|dqacc| %.2e" % np.abs(b.field(S.F_QACC)[0].cpu().numpy() - o.get("qacc")).max())
act = torch.zeros(n, ctrl.action_dim, dtype=torch.float64, device="cuda"); tb = torch.zeros(n, 69, dtype=torch.float64, device="cuda")
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    b.simulate(act, tb); b.sync()
    o.do_simulation(np.zeros(ctrl.action_dim), np.zeros(69))
    print(f"substep {t}: nefc", b.field(S.F_NEFC).tolist(), "iters", b.field(S.F_SOLVER_ITER).tolist(), "redo", [hex(int(x)) for x in b.field(S.F_REDO).tolist()], "fail", b.field(S.F_FAIL).tolist(),
          "oracle nefc", o.geti("nefc"), "iters", o.geti("solver_iter"), "|dqpos| %.2e" % np.abs(b.field(S.F_QPOS)[0].cpu().numpy() - o.get("qpos")).max(),
          "|dqacc| %.2e" % np.abs(b.field(S.F_QACC)[0].cpu().numpy() - o.get("qacc")).max(), "|qacc| %.2e" % np.abs(o.get("qacc")).max())
