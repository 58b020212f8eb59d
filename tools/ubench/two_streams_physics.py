"""Do two fused step kernels on two HIP streams run concurrently (512 envs each = half the SIMD slots each)?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from uhc_amd import sim as S
model = S.load_asset_model(); ctrl = S.make_ctrl(model)
z = np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz"))
def mk(n, stream):
    with torch.cuda.stream(stream):
        b = S.SimBatch(model, ctrl, n)
        rng = np.random.default_rng(1)
        qpos = np.tile(z["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(scale=0.05, size=(n, model.nu))
        b.set_state(torch.from_numpy(qpos), torch.from_numpy(rng.normal(scale=0.1, size=(n, model.nv))))
        a = torch.from_numpy(rng.normal(scale=0.1, size=(n, ctrl.action_dim))).cuda()
        tb = torch.from_numpy(np.ascontiguousarray(qpos[:, 7:])).cuda()
    return b, a, tb
for G, n in ((1, 1024), (2, 512), (1, 512), (2, 1024)):
    streams = [torch.cuda.Stream() for _ in range(G)]
    bs = [mk(n, s) for s in streams]
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(10):
            for (b, a, tb), s in zip(bs, streams):
                with torch.cuda.stream(s):
                    b.simulate(a, tb)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 10
    print(f"{G} stream(s) x {n} envs: {el * 1e3:.2f} ms per round, {G * n / el:.0f} env-steps/s")
    for b, _, _ in bs: b.close()
