"""GPU diagnostic: per control step of a bench.py rollout probe, read UHC_F_REDO back and tabulate (a) how the exact contact solve of the
general / large tier ended -- direct, in windows of 64 rows (bit 3), or by the sweeps fallback with its reason (bits 2, 4, 5) -- and (b) the
wall time of the steps in which a fallback / a windowed solve happened against the others.

  python tools/diag_redo.py configs4 [steps] [warmup] [envs]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "configs4"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    envs = int(sys.argv[4]) if len(sys.argv) > 4 else None
    from uhc_amd import sim as S
    sys.argv = sys.argv[:1]
    args = bench.parse()
    torch.cuda.set_device(0)
    torch.set_default_dtype(torch.float64)
    kw = {k: v for k, v in bench.PROBES[name].items() if k not in ("name", "steps")}
    if envs:
        kw["envs"] = envs
    agent = bench.build_agent(args, 0, 0, torch.float64, **kw)
    agent.per_epoch_update(0)
    env = agent.env
    agent.rollout_begin(warmup + steps)
    for _ in range(warmup):
        agent.rollout_step()
    torch.cuda.synchronize()
    rows = []
    dump, cases = os.environ.get("DIAG_DUMP"), []  # DIAG_DUMP=file.pkl: the start state, action and UHC_F_REDO of every env-step that fell back (CPU replay)
    for t in range(steps):
        if dump:
            q0, v0 = env.sim.field(S.F_QPOS).clone(), env.sim.field(S.F_QVEL).clone()
        t0 = time.perf_counter()
        agent.rollout_step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
        redo = env.sim.field(S.F_REDO).cpu().numpy()
        nefc = env.sim.field(S.F_NEFC).cpu().numpy()
        if dump and len(cases) < 60:
            for e in np.nonzero((redo & 2) != 0)[0]:
                cases.append(dict(step=t, env=int(e), redo=int(redo[e]), qpos=q0[e].cpu().numpy(), qvel=v0[e].cpu().numpy(), action=agent._ro.action[e].cpu().numpy(), nefc=int(nefc[e])))
        rows.append((ms, int(((redo & 2) != 0).sum()), int(((redo & 8) != 0).sum()), int(((redo & 4) != 0).sum()), int(((redo & 16) != 0).sum()),
                     int(((redo & 32) != 0).sum()), int(((redo & 0x40) != 0).sum()), int(nefc[(redo & 2) != 0].max(initial=0)),
                     int(np.unpackbits(((redo[(redo & 2) != 0] >> 8) & 0x7fff).astype(">u2").view(np.uint8)).sum())))
    agent.rollout_end()
    if dump:
        import pickle
        pickle.dump(dict(model=env.model, cases=cases), open(dump, "wb"))
        print(f"  {len(cases)} fallback env-steps written to {dump}")
    a = np.array(rows, dtype=float)
    n_env = env.n_env
    print(f"{name}: {steps} control steps x {n_env} envs after {warmup} warm-up steps (each step synchronised: wall times include the launch latency)")
    print(f"  env-steps whose exact solve fell back to sweeps: {int(a[:, 1].sum())} (friction-loss rows {int(a[:, 3].sum())}, no convergence of the working sets "
          f"{int(a[:, 4].sum())}, pivoting gave up on a working set {int(a[:, 5].sum())}); substeps swept in them: {int(a[:, 8].sum())}; "
          f"env-steps with a windowed exact solve: {int(a[:, 2].sum())}; large-tier env-steps {int(a[:, 6].sum())}")
    fb, wn = a[:, 1] > 0, (a[:, 2] > 0) & (a[:, 1] == 0)
    rest = ~fb & ~wn
    for lab, sel in (("steps with a sweeps fallback", fb), ("steps with a windowed solve, no fallback", wn), ("other steps", rest)):
        if sel.any():
            print(f"  {lab}: {int(sel.sum())}; wall ms (median / p90 / max) {np.median(a[sel, 0]):.2f} / {np.percentile(a[sel, 0], 90):.2f} / {a[sel, 0].max():.2f}")
    print(f"  all steps: mean {a[:, 0].mean():.2f} ms, median {np.median(a[:, 0]):.2f} ms; the ten slowest: " +
          ", ".join(f"{r[0]:.1f} ms (fallback {int(r[1])}, windowed {int(r[2])}, nefc {int(r[7])})" for r in sorted(rows, key=lambda r: -r[0])[:10]))


if __name__ == "__main__":
    main()
