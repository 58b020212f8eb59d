"""Size-independent properties at BASELINE.json's full batch sizes (configs[1]: 1024 envs / GPU, configs[2]: 4096 envs / GPU).
The oracle cannot follow these sizes in seconds; what is checked instead does not need it:
  * an env's result does not depend on the batch it runs in (bit-identical to the same env inside a 16-env batch -- the size at
    which the parity tests pin the kernel against the oracle), nor on the split of the batch across ranks;
  * the step is a pure function of its inputs (same state + action twice -> same bits), forward() is idempotent;
  * without contact the centre of mass falls with g whatever the joint torques do;
  * the active-set contact solve ends within a handful of factorisations everywhere, nothing fails, few envs need the general kernel."""
import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _states(model, standing, n, seed, lift=0.0):
    rng = np.random.default_rng(seed)
    qpos = np.tile(standing["qpos"], (n, 1))
    qpos[:, 7:] += rng.normal(scale=0.05, size=(n, model.nu))
    qpos[:, 2] += lift
    qvel = rng.normal(scale=0.1, size=(n, model.nv))
    return qpos, qvel


def _run(model, ctrl, qpos, qvel, actions, fields, steps):
    import torch
    from uhc_amd import sim as S
    n = qpos.shape[0]
    b = S.SimBatch(model, ctrl, n)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    tb = torch.from_numpy(np.ascontiguousarray(qpos[:, 7:])).cuda()
    a = torch.from_numpy(actions).cuda()
    for t in range(steps):
        b.simulate(a[t], tb)
    b.sync()
    out = {k: b.field(f).cpu().numpy().copy() for k, f in fields.items()}
    b.close()
    return out


@pytest.mark.parametrize("n_env", [1024, 4096])
@pytest.mark.parametrize("solver", [1, 0])
def test_env_results_do_not_depend_on_the_batch(model, ctrl, standing, n_env, solver):
    from uhc_amd import sim as S
    model = dataclasses.replace(model, solver=solver, iterations=100)
    steps = 6
    qpos, qvel = _states(model, standing, n_env, 100 + n_env)
    actions = np.random.default_rng(7).normal(scale=np.exp(-2.3), size=(steps, n_env, ctrl.action_dim))
    fields = dict(qpos=S.F_QPOS, qvel=S.F_QVEL, qacc=S.F_QACC, nefc=S.F_NEFC, it=S.F_SOLVER_ITER, fail=S.F_FAIL, redo=S.F_REDO, ov=S.F_EFC_OVERFLOW)
    full = _run(model, ctrl, qpos, qvel, actions, fields, steps)
    assert not full["fail"].any() and not full["ov"].any() and np.isfinite(full["qpos"]).all() and np.isfinite(full["qvel"]).all()
    assert full["nefc"].max() <= 64 or full["redo"].any()
    if solver == 1:  # a handful of factorisations everywhere (an env of the general kernel reports sweeps instead)
        fast = full["redo"] == 0
        assert full["it"][fast].max() <= 12 and fast.mean() > 0.9
    # (a) the same again: bit-identical
    again = _run(model, ctrl, qpos, qvel, actions, fields, steps)
    for k in ("qpos", "qvel", "qacc"):
        assert np.array_equal(full[k], again[k]), k
    # (b) 16 envs picked from the batch, alone in a 16-env batch: bit-identical (one env per wavefront, no cross-env arithmetic)
    pick = np.random.default_rng(3).choice(n_env, 16, replace=False)
    small = _run(model, ctrl, qpos[pick], qvel[pick], np.ascontiguousarray(actions[:, pick]), fields, steps)
    for k in ("qpos", "qvel", "qacc", "nefc"):
        assert np.array_equal(full[k][pick], small[k]), k
    # (c) the batch split across two "ranks" (env sharding of the multi-GPU path): the halves reproduce the whole
    h = n_env // 2
    lo = _run(model, ctrl, qpos[:h], qvel[:h], np.ascontiguousarray(actions[:, :h]), fields, steps)
    hi = _run(model, ctrl, qpos[h:], qvel[h:], np.ascontiguousarray(actions[:, h:]), fields, steps)
    assert np.array_equal(np.concatenate([lo["qpos"], hi["qpos"]]), full["qpos"])


def test_forward_is_idempotent_and_free_fall_keeps_g(model, ctrl, standing):
    """1024 airborne humanoids with random joint torques: no contact rows, and the centre of mass of every env follows
    x0 + v0 t - g t^2 / 2 (internal forces cannot move it); forward() on an unchanged state changes nothing."""
    import torch
    from uhc_amd import sim as S
    n, steps = 1024, 6
    qpos, qvel = _states(model, standing, n, 5, lift=1.5)
    qvel[:, 3:6] = 0.0  # no spin: keeps the generalised-coordinate integration error of the COM small
    b = S.SimBatch(model, ctrl, n)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    b.sync()
    mass = np.asarray(model.body_mass)[None, :, None]
    com = lambda: (b.field(S.F_XIPOS).cpu().numpy().reshape(n, -1, 3) * mass).sum(1) / mass.sum()
    snap = {k: b.field(f).clone() for k, f in dict(x=S.F_XPOS, q=S.F_XQUAT, a=S.F_QACC, m=S.F_QM).items()}
    b.forward()
    b.sync()
    for k, f in dict(x=S.F_XPOS, q=S.F_XQUAT, a=S.F_QACC, m=S.F_QM).items():
        assert torch.equal(snap[k], b.field(f)), k
    c = []  # the body fields are those of the last forward pass, i.e. one substep before the step's end: uniform spacing from step 1 on
    tb = torch.from_numpy(np.ascontiguousarray(qpos[:, 7:])).cuda()
    # PD torques towards the initial pose only (zero action, no residual root wrench): internal forces, gentle enough for the
    # joint-space semi-implicit integration to keep the COM's discrete acceleration within 1e-3 of g
    act = np.zeros((steps, n, ctrl.action_dim))
    for t in range(steps):
        b.simulate(torch.from_numpy(act[t]).cuda(), tb)
        b.sync()
        assert int(b.field(S.F_NEFC).max().item()) <= 69  # joint limits at most, no contact rows in the air
        c.append(com())
    c = np.array(c)
    dt = model.timestep * 15
    acc = (c[2:] - 2 * c[1:-1] + c[:-2]) / dt ** 2   # second difference of the COM path per control step
    print(f"COM acceleration error: horizontal {np.abs(acc[..., :2]).max():.2e}, vertical {np.abs(acc[..., 2] + 9.81).max():.2e}")
    assert np.abs(acc[..., :2]).max() < 1e-3 and np.abs(acc[..., 2] + 9.81).max() < 1e-3  # measured 1.2e-4 (discretisation of the PD-driven joints)
    b.close()


def test_self_colliding_batch_does_not_depend_on_the_fast_tier_layout(model, ctrl, standing):
    """A batch of >= 3072 self-colliding humanoids gets the fast tier's 40 KiB / 6 body-body-row layout (four workgroups per CU), a smaller
    one the 52 KiB / 12-row layout: more envs of the big batch are handed to the general tier.  Every tier solves the same QP exactly, so
    an env of the 3072-env batch agrees with the same env inside a 16-env batch to rounding -- not to the bit: another tier sums in another
    order -- and a rerun of the big batch is bit-identical."""
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import self_collision_variant
    m = dataclasses.replace(self_collision_variant(model), solver=1)
    n_env, steps = 3072, 6
    qpos, qvel = _states(m, standing, n_env, 4242)
    actions = np.random.default_rng(8).normal(scale=np.exp(-2.3), size=(steps, n_env, ctrl.action_dim))
    fields = dict(qpos=S.F_QPOS, qvel=S.F_QVEL, nefc=S.F_NEFC, fail=S.F_FAIL, redo=S.F_REDO, ov=S.F_EFC_OVERFLOW)
    full = _run(m, ctrl, qpos, qvel, actions, fields, steps)
    assert not full["fail"].any() and not full["ov"].any() and not (full["redo"] & 2).any()
    again = _run(m, ctrl, qpos, qvel, actions, fields, steps)
    assert np.array_equal(full["qpos"], again["qpos"]) and np.array_equal(full["qvel"], again["qvel"])
    pick = np.random.default_rng(3).choice(n_env, 16, replace=False)
    small = _run(m, ctrl, qpos[pick], qvel[pick], np.ascontiguousarray(actions[:, pick]), fields, steps)
    moved = int(((full["redo"][pick] & 1) != (small["redo"] & 1)).sum())
    dq, dv = np.abs(full["qpos"][pick] - small["qpos"]).max(), np.abs(full["qvel"][pick] - small["qvel"]).max()
    print(f"3072-env batch vs 16-env batch: {moved} of 16 envs ended in another tier; |dqpos| {dq:.2e} |dqvel| {dv:.2e}; general-tier share of the big batch {(full['redo'] & 1).mean():.3f}")
    assert np.array_equal(full["nefc"][pick], small["nefc"])
    assert dq < 1e-9 and dv < 1e-7, (dq, dv)
