"""GPU diagnostic (instrumented library, tools/stage_profile.py --build-only): which stages make the SLOWEST general-tier env of a control step
slow.  Runs a bench.py rollout (the headline, or a probe by key) on libuhc_amd_prof.so, reads the per-env stage cycle counters
(UHC_F_STAGE_PROF) after every step and prints, per stage, the mean over the general-tier envs and the mean over each step's slowest one.

  python tools/diag_slowest.py [probe key | headline] [steps] [warmup]
  SELECT=tier4: the envs whose step went through tier 4 (UHC_F_REDO bit 30) instead of all general / large-tier envs
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("UHC_LIB", os.path.join(ROOT, "uhc_amd", "csrc", "libuhc_amd_prof.so"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from stage_profile import NAMES  # noqa: E402


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "headline"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    from uhc_amd import sim as S
    sys.argv = sys.argv[:1]
    args = bench.parse()
    torch.cuda.set_device(0)
    torch.set_default_dtype(torch.float64)
    kw = {} if key == "headline" else {k: v for k, v in bench.PROBES[key].items() if k not in ("name", "steps")}
    agent = bench.build_agent(args, 0, 0, torch.float64, **kw)
    agent.per_epoch_update(0)
    env = agent.env
    agent.rollout_begin(warmup + steps)
    for _ in range(warmup):
        agent.rollout_step()
    torch.cuda.synchronize()
    prof = env.sim.field(S.F_STAGE_PROF)
    n = env.n_env
    acc_all, acc_slow, n_all, tot_slow, tot_all, tot_fast, nefc_slow = np.zeros(40), np.zeros(40), 0, [], [], [], []
    for _ in range(steps):
        prof.zero_()
        agent.rollout_step()
        torch.cuda.synchronize()
        p = prof.cpu().numpy().reshape(n, 40).astype(np.float64)
        redo = env.sim.field(S.F_REDO).cpu().numpy()
        nefc = env.sim.field(S.F_NEFC).cpu().numpy()
        top = p[:, :16].sum(1)
        g = np.nonzero((redo & (1 << 30)) != 0)[0] if os.environ.get("SELECT") == "tier4" else np.nonzero((redo & 1) != 0)[0]
        if len(g) == 0:
            continue
        e = g[np.argmax(top[g])]
        acc_all += p[g].sum(0)
        n_all += len(g)
        acc_slow += p[e]
        tot_slow.append(top[e]); tot_all.append(top[g].mean()); tot_fast.append(top[(redo & 1) == 0].mean()); nefc_slow.append(int(nefc[e]))
        if os.environ.get("SELECT") == "tier4":
            print(f"  step: {len(g)} envs through tier 4; cycles {sorted(int(x) for x in top[g])}; nefc {[int(nefc[x]) for x in g]}; substeps swept/resumed word {[hex(int(redo[x])) for x in g][:4]}", flush=True)
    agent.rollout_end()
    k = len(tot_slow)
    print(f"{key}: {k} control steps x {n} envs after {warmup} warm-up steps; cycles per env-step in the top-level stages: fast-tier envs {np.mean(tot_fast):.3g}, "
          f"general / large-tier envs {np.mean(tot_all):.3g} ({n_all / k:.1f} per step), the slowest of them per step {np.mean(tot_slow):.3g} (max {np.max(tot_slow):.3g}); "
          f"nefc at the end of the slowest envs' steps {sorted(nefc_slow)[:3]} ... {sorted(nefc_slow)[-3:]}")
    print(f"  {'stage':36s} {'mean of the tier':>16s} {'slowest per step':>16s}  ratio")
    for i, name in enumerate(NAMES):
        a, b = acc_all[i] / max(n_all, 1), acc_slow[i] / k
        if a > 0 or b > 0:
            print(f"  {name[:36]:36s} {a:16.0f} {b:16.0f}  x{b / max(a, 1):.2f}")


if __name__ == "__main__":
    main()
