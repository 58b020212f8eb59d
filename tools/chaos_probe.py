"""How fast do two copies of the SAME solver drift apart?  The oracle against itself on the 200-step parity workloads (tests/test_gpu_parity_200.py), the
second copy started with qpos perturbed by 1e-14 (relative to one unit in the last place of a joint angle): per workload and env, the first step at which the two trajectories differ by more than 1e-12 / 1e-10 / 1e-8 /
1e-6 / 1e-4.  What the device-vs-oracle comparison can hold over 200 free-running steps is bounded by this: the two implementations sum in different
orders, i.e. differ by rounding from the first step on.   python tools/chaos_probe.py > profiles/r05_chaos_probe.txt"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.physics import OracleSim  # noqa: E402
from tests.test_gpu_parity_200 import _class  # noqa: E402
from uhc_amd.sim import load_asset_model  # noqa: E402

model = load_asset_model()
z = np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz"))
standing = {k: z[k] for k in z.files}
print(__doc__)
for name in ("generated", "shape", "ball", "ball_objects"):
    m, ctrl, q0, v0, a_sc, tb = _class(name, model, standing)
    n = q0.shape[0]
    A = [OracleSim(m, ctrl) for _ in range(n)]
    B = [OracleSim(m, ctrl) for _ in range(n)]
    prng = np.random.default_rng(1)
    for e in range(n):
        A[e].set_state(q0[e], v0[e])
        qb = q0[e].copy()
        qb[7:] *= 1.0 + 1e-14 * prng.choice([-1.0, 1.0], size=qb[7:].shape)
        B[e].set_state(qb, v0[e])
    rng = np.random.default_rng(7)
    err = np.zeros((200, n))
    for t in range(200):
        act = rng.normal(scale=a_sc, size=(n, ctrl.action_dim))
        for e in range(n):
            A[e].do_simulation(act[e], tb[e])
            B[e].do_simulation(act[e], tb[e])
            err[t, e] = max(np.abs(A[e].get("qpos") - B[e].get("qpos")).max(), np.abs(A[e].get("qvel") - B[e].get("qvel")).max())
    for e in range(n):
        first = {f"1e{k}": (int(np.nonzero(err[:, e] > 10.0 ** k)[0][0]) if (err[:, e] > 10.0 ** k).any() else None) for k in (-12, -10, -8, -6, -4)}
        print(f"{name:13s} env {e}: max over 200 steps {err[:, e].max():.2e}; first step above {first}")
