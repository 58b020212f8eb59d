"""CPU study (oracle only, test infrastructure): how many rounds does a windowed block solve of the contact QP need when one island carries a
force on more than 64 rows?  The kernel's working-set solver (uhc_physics_impl.h, k_as_general) solves at most 64 rows at once exactly; this
script replays the scheme it uses beyond that -- the rows of the island that are not in the current window keep their force, the window's
sub-QP is solved exactly against them, the window moves on cyclically -- on Delassus matrices taken from oracle roll-outs of the
configs[4] stand-in scene (ball-joint humanoid, body-body collisions, free boxes, random torques), and prints rounds to the KKT tolerance.

  python tools/proto_block_cd.py [--envs 4] [--steps 70] [--objects 4]
"""
import argparse
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def exact_subqp(A, b, f0=None):
    """min 1/2 f'Af + f'b, f >= 0, by block principal pivoting with a single-index fallback (the oracle's rule)."""
    n = len(b)
    F = np.zeros(n, bool) if f0 is None else f0 > 0
    best, grace = n + 1, 3
    f = np.zeros(n)
    for it in range(200):
        f[:] = 0
        if F.any():
            f[F] = np.linalg.solve(A[np.ix_(F, F)], -b[F])
        y = A @ f + b
        bad = np.where(F, f < 0, y < 0)
        nb = int(bad.sum())
        if nb == 0:
            return f, it + 1
        if nb < best:
            best, grace = nb, 3
            F[bad] = ~F[bad]
        elif grace > 0:
            grace -= 1
            F[bad] = ~F[bad]
        else:
            k = np.nonzero(bad)[0][-1]
            F[k] = ~F[k]
    raise RuntimeError("pivoting did not converge")


def islands(A):
    n = A.shape[0]
    lab = np.arange(n)
    nz = np.abs(A) > 1e-14
    changed = True
    while changed:
        changed = False
        for r in range(n):
            m = lab[nz[r]].min()
            if m < lab[r]:
                lab[r] = m
                changed = True
            sel = nz[r] & (lab > m)
            if sel.any():
                lab[sel] = m
                changed = True
    return lab


def block_cd(A, b, f_warm, W=64, tol_rel=1e-9, maxit=400, overlap_policy="cyclic"):
    """Windowed block coordinate descent as the kernel would run it on ONE island.  Returns rounds, final KKT residual."""
    n = len(b)
    f = np.maximum(f_warm, 0.0).copy()
    tol = tol_rel * max(1.0, np.abs(b).max())
    cursor = 0
    y = A @ f + b
    for rnd in range(maxit):
        interesting = (f > 0) | (y < -tol)
        idx = np.nonzero(interesting)[0]
        if len(idx) == 0:
            return rnd, 0.0
        if len(idx) <= W:
            C = idx
        else:
            order = np.r_[idx[idx >= cursor], idx[idx < cursor]]
            C = np.sort(order[:W])
            cursor = (order[W - 1] + 1) % n
        fixed = np.ones(n, bool)
        fixed[C] = False
        bb = b[C] + A[np.ix_(C, fixed)] @ f[fixed]
        fc, _ = exact_subqp(A[np.ix_(C, C)], bb)
        f[C] = fc
        y = A @ f + b
        res = max((-y[f == 0]).max(initial=0.0), np.abs(y[f > 0]).max(initial=0.0))
        if res <= tol:
            return rnd + 1, res
    return maxit, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=3)
    ap.add_argument("--steps", type=int, default=70)
    ap.add_argument("--objects", type=int, default=4)
    ap.add_argument("--window", type=int, default=64)
    args = ap.parse_args()
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    from uhc_amd.model.shapes import box_triangles
    from uhc_amd.model.mjcf import add_free_bodies, ball_variant, hinge_to_ball_qpos, self_collision_variant
    base = S.load_asset_model()
    rng = np.random.default_rng(11)
    K = args.objects
    ang = rng.uniform(0, 2 * np.pi, size=K)
    poses = np.stack([np.r_[-0.15 + 0.75 * np.cos(a), -0.05 + 0.75 * np.sin(a), 0.3 + 0.45 * k, 1, 0, 0, 0] for k, a in enumerate(ang)])
    hb = ball_variant(base, damping=5.0)
    m = self_collision_variant(hb)
    m = add_free_bodies(m, [box_triangles(0.15, 0.15, 0.15)] * K, poses, density=5.0 / 0.027)
    m = dataclasses.replace(m, solver=1, iterations=300)
    ctrl = S.make_ctrl(base, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
    stand = np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz"))["qpos"]
    tb = np.zeros(69)
    rounds_all = []
    for e in range(args.envs):
        o = OracleSim(m, ctrl)
        q = m.qpos0.copy()
        qh = stand.copy()
        qh[7:] += rng.normal(scale=0.1, size=69)
        q[:99] = hinge_to_ball_qpos(base, hb, qh)
        v = np.zeros(m.nv)
        v[:75] = rng.normal(scale=0.2, size=75)
        o.set_state(q, v)
        for t in range(args.steps):
            act = 0.1 * rng.normal(size=ctrl.action_dim)
            fprev = None
            o.do_simulation(act, tb)
            if o.geti("fail"):
                print(f"  env {e} step {t}: oracle reports a bad state, next env", flush=True)
                break
            o.forward()
            n = o.geti("nefc")
            if n == 0:
                continue
            A = o.get("efc_AR").reshape(n, n)
            b = o.get("efc_b")
            f = o.get("efc_force")
            if t % 10 == 0:
                print(f"  env {e} step {t}: nefc {n} with force {int((f > 0).sum())} root z {o.get('qpos')[2]:.2f}", flush=True)
            lab = islands(A)
            for I in np.unique(lab):
                rows = np.nonzero(lab == I)[0]
                nact = int((f[rows] > 0).sum())
                if nact <= args.window - 8:
                    continue
                AI, bI = A[np.ix_(rows, rows)], b[rows]
                # warm start the way the kernel has it: the previous forces, perturbed (contacts persist between substeps)
                fw = f[rows] * (1 + 0.2 * rng.normal(size=len(rows)))
                r_warm, res_w = block_cd(AI, bI, fw, W=args.window)
                r_cold, res_c = block_cd(AI, bI, np.zeros(len(rows)), W=args.window)
                rounds_all.append((len(rows), nact, r_warm, r_cold))
                print(f"env {e} step {t}: nefc {n} island rows {len(rows)} with force {nact}: rounds warm {r_warm} (res {res_w:.1e}) cold {r_cold} (res {res_c:.1e})", flush=True)
    if rounds_all:
        a = np.array(rounds_all)
        big = a[a[:, 1] > args.window]
        print(f"{len(a)} islands near / over the window; over: {len(big)}; rounds warm (median / p90 / max) "
              f"{np.median(big[:, 2]) if len(big) else 0} / {np.percentile(big[:, 2], 90) if len(big) else 0} / {big[:, 2].max() if len(big) else 0}; "
              f"cold {np.median(big[:, 3]) if len(big) else 0} / {np.percentile(big[:, 3], 90) if len(big) else 0} / {big[:, 3].max() if len(big) else 0}")


if __name__ == "__main__":
    main()
