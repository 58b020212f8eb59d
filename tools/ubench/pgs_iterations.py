import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from uhc_amd import sim as S
model = S.load_asset_model()
ctrl = S.make_ctrl(model)
z = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "uhc_amd", "assets", "standing_neutral.npz"))
n_env = 1024
rng = np.random.default_rng(1)
for iters in (100, 200, 400):
    model.iterations = iters
    b = S.SimBatch(model, ctrl, n_env)
    qpos = np.tile(z["qpos"], (n_env, 1)); qpos[:, 7:] += rng.normal(scale=0.05, size=(n_env, model.nu))
    qvel = rng.normal(scale=0.1, size=(n_env, model.nv))
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    a = torch.from_numpy(rng.normal(scale=np.exp(-2.3), size=(8, n_env, ctrl.action_dim))).cuda()
    tb = torch.from_numpy(np.ascontiguousarray(qpos[:, 7:])).cuda()
    for t in range(3): b.simulate(a[t % 8], tb)
    b.sync(); t0 = time.perf_counter()
    for t in range(10): b.simulate(a[t % 8], tb)
    b.sync(); el = (time.perf_counter() - t0) / 10
    it = b.field(S.F_SOLVER_ITER).float()
    print(f"cap {iters}: {el*1e3:.2f} ms/launch, {n_env/el:.0f} env-steps/s, last-substep sweeps mean {it.mean().item():.1f} max {it.max().item():.0f}, at cap {(it >= iters).float().mean().item()*100:.0f}%")
    b.close()
