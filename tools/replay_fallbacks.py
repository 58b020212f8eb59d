"""CPU replay (oracle only, test infrastructure) of the env-steps a GPU run wrote with `DIAG_DUMP=file.pkl python tools/diag_redo.py <probe>`:
every env-step whose exact contact solve fell back to the sweeps, with its start state, action and UHC_F_REDO word.  The oracle steps
from the start state to the first swept substep and prints what the solver faced there: rows (the oracle is not capped at 256), rows with a
force, the largest |b|, the joint speeds at the head of the step, and how the device's working-set scheme (emulated in numpy: working sets of
<= 64 rows, block pivoting with the single-index rule after three rounds without progress, 64 rounds) ends on the biggest island.

  python tools/replay_fallbacks.py gpurun_out/fallback_cases.pkl [n_cases]
"""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from proto_block_cd import islands  # noqa: E402


def as_solve(A, b, maxit=64, presweeps=16):
    n = len(b)
    f = np.zeros(n)
    for _ in range(presweeps):
        for r in range(n):
            f[r] = max(0.0, f[r] - (b[r] + A[r] @ f) / A[r, r])
    F = f > 0
    best, grace, hist = n + 1, 3, []
    for it in range(1, maxit + 1):
        f = np.zeros(n)
        if F.any():
            f[F] = np.linalg.solve(A[np.ix_(F, F)], -b[F])
        y = b + A @ f
        bad = np.where(F, f < 0, y < 0)
        nb = int(bad.sum())
        hist.append(nb)
        if nb == 0:
            return f, it, hist
        allf = True
        if nb < best:
            best, grace = nb, 3
        elif grace > 0:
            grace -= 1
        else:
            allf = False
        if allf:
            F[bad] = ~F[bad]
        else:
            k = np.nonzero(bad)[0][-1]
            F[k] = ~F[k]
    return None, maxit, hist


def ws_solve(A, b, fwarm):
    """the device's working-set loop on one island, up to the point where it would go over to windows"""
    n = len(b)
    c, p = fwarm > 0, np.zeros(n, bool)
    for outer in range(16):
        idx = np.nonzero(c)[0]
        if len(idx) > 64:
            q = c & ~p
            room = 64 - (c.sum() - q.sum())
            if room <= 0:
                return f"64 rows carry a force and more want in after {outer} rounds: windows"
            c = p.copy()
            c[np.nonzero(q)[0][:room]] = True
            idx = np.nonzero(c)[0]
        if len(idx) == 0:
            c = b < 0
            if not c.any():
                return "solved"
            continue
        fc, it, hist = as_solve(A[np.ix_(idx, idx)], b[idx])
        if fc is None:
            return f"the pivoting gave up on a working set of {len(idx)} rows (rows with the wrong sign per round: {hist[:8]} ... {hist[-4:]})"
        f = np.zeros(n)
        f[idx] = fc
        y = b + A @ f
        viol, keep = (~c) & (y < 0), c & (f > 0)
        if not viol.any():
            return f"solved in {outer + 1} working sets"
        c, p = keep | viol, keep
    return "no convergence in 16 working sets"


def main():
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    D = pickle.load(open(sys.argv[1], "rb"))
    model = D["model"]
    base = S.load_asset_model()
    ctrl = S.make_ctrl(base, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)  # (the ball-joint probes' controller)
    nu = model.nu
    a_scale = np.ctypeslib.as_array(ctrl.a_scale, (nu,)).copy()
    lim = np.ctypeslib.as_array(ctrl.torque_lim, (nu,)).copy()
    for case in D["cases"][:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
        o = OracleSim(model, ctrl)
        o.set_state(case["qpos"], case["qvel"])
        mask = (case["redo"] >> 8) & 0x7fff
        ks = [k for k in range(15) if (mask >> k) & 1]
        o.set("ctrl", np.clip(case["action"][:nu] * a_scale * 100, -lim, lim))
        for _ in range(ks[0]):
            o.step()
        o.forward()
        n = o.geti("nefc")
        A, b, f = o.get("efc_AR").reshape(n, n), o.get("efc_b"), o.get("efc_force")
        lab = islands(A)
        big = max(np.unique(lab), key=lambda I: (lab == I).sum())
        rows = np.nonzero(lab == big)[0]
        fw = f[rows] * (1 + 0.1 * np.random.default_rng(0).normal(size=len(rows)))  # stand-in for the device's warm start
        print(f"step {case['step']} env {case['env']} UHC_F_REDO {case['redo']:#x} (rows lost: {bool(case['redo'] & 0x80)}) first swept substep {ks[0]}: "
              f"max |qvel| at the head of the step {np.abs(case['qvel']).max():.3g}; at the substep: {n} rows, {int((f > 0).sum())} with a force, "
              f"max |b| {np.abs(b).max():.3g}, oracle's own active set {o.geti('solver_iter')} rounds; biggest island {len(rows)} rows: {ws_solve(A[np.ix_(rows, rows)], b[rows], fw)}", flush=True)


if __name__ == "__main__":
    main()
