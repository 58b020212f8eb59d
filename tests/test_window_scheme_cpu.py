"""The window scheme of the general tier's exact contact solve (uhc_physics_impl.h, k_as_general: islands with more than 64 force-carrying rows
are solved in windows of 64 rows, the rows outside a window hold their force) restated in numpy (tools/proto_block_cd.py) and checked
on CPU against a solve of all rows at once: it converges to the same optimum, monotonically, from a warm and from a cold start."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _problem(n, nv, seed):
    """A = J J' + R (sparse J, a small regularisation R as the soft-contact model leaves it), b so that most rows carry a force."""
    rng = np.random.default_rng(seed)
    J = rng.normal(size=(n, nv)) * (rng.random((n, nv)) < 0.3)
    A = J @ J.T + np.diag(rng.uniform(1e-3, 1e-1, size=n))
    b = rng.normal(size=n) - 1.5
    return A, b


def test_windows_reach_the_optimum_of_the_whole_problem():
    from proto_block_cd import block_cd, exact_subqp
    for seed, n in ((1, 96), (2, 130), (3, 200)):
        A, b = _problem(n, 120, seed)
        f_all, _ = exact_subqp(A, b)
        assert (f_all > 0).sum() > 64, "the case must need windows"
        y = A @ f_all + b
        assert (y[f_all == 0] >= -1e-9).all() and np.abs(y[f_all > 0]).max() < 1e-8  # KKT of the reference solution
        for start in (np.zeros(n), np.maximum(f_all * (1 + 0.3 * np.random.default_rng(seed).normal(size=n)), 0)):
            rounds, res = block_cd(A, b, start, W=64, tol_rel=1e-8, maxit=2000)  # (random matrices of this size stall in rounding just below 1e-8)
            assert rounds < 2000 and res <= 1e-8 * (1 + np.abs(b).max()), (seed, rounds, res)


def test_every_window_lowers_the_dual_cost():
    """One window solve minimises 1/2 f'Af + f'b over its rows with the others held: the cost never rises (what makes the iteration converge)."""
    from proto_block_cd import exact_subqp
    A, b = _problem(150, 50, 7)
    n = len(b)
    f = np.zeros(n)
    cost = lambda f: 0.5 * f @ A @ f + f @ b
    last = cost(f)
    cursor = 0
    for _ in range(40):
        y = A @ f + b
        cand = np.nonzero((f > 0) | (y < 0))[0]
        if len(cand) == 0:
            break
        order = np.r_[cand[cand >= cursor], cand[cand < cursor]]
        C = np.sort(order[:64])
        cursor = (order[min(63, len(order) - 1)] + 1) % n
        held = np.ones(n, bool)
        held[C] = False
        f[C], _ = exact_subqp(A[np.ix_(C, C)], b[C] + A[np.ix_(C, held)] @ f[held])
        now = cost(f)
        assert now <= last + 1e-9 * max(1.0, abs(last)), (now, last)
        last = now
