"""north_star's bar on the workloads bench.py reports (VERDICT r4 "next" 1): per-step qpos / qvel of the HIP path within 1e-4 of the CPU
oracle over 200 control steps, same seed -- on the generated model class (body-body collisions + rel_joint_lm ranges: the headline), one
per-body-scaled shape of it (configs[3]), the ball-joint humanoid (configs[4]'s env without objects) and the ball-joint humanoid among four
boxes (configs[4]) -- through the tier chain (fast first), the general tier alone and the sticky queues, each with the device's per-substep
solver word handed to the oracle AND with nothing handed over (the oracle decides its own solver: a wrong fallback decision on the device
would show).  A trajectory that leaves 1e-4 is reported with the step it leaves at and what happened there (rows dropped, sweeps fallback,
contact-set flip, or none of them = rounding amplified by the dynamics); every run appends its record to gpurun_out/parity200.jsonl.

What round 5 found (profiles/r05_*_parity200*.jsonl, profiles/r05_chaos_probe.txt): the hinge classes with the PD controller hold 1e-11 (generated) /
5e-10 (shape) over all 200 steps in every mode.  The ball-joint classes do not, and cannot: a ball-joint humanoid without joint limits, damping or PD
(copycat_ball_1.yml drives torques) lying on the floor is a chain of free pendulums -- the ORACLE AGAINST ITSELF, started 1e-14 apart, leaves 1e-4
after 57-72 control steps (tools/chaos_probe.py), the device against the oracle after 57-73, decade by decade at the same steps.  Their bar is
therefore (a) the free run: 1e-7 over the first 25 steps, the exit reported; (b) the same 200 steps with the oracle re-started from the device's
state at every step: each single control step within 1e-9 -- parity along the whole trajectory without the dynamics' own amplification."""
import dataclasses
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_STEPS = 200
TOL = 1e-4


def _class(name, model, standing):
    """-> (model, ctrl, qpos0 [n, nq], qvel0 [n, nv], action scale, target_base [n, 69])"""
    from uhc_amd.model.mjcf import add_free_bodies, ball_variant, hinge_to_ball_qpos, kinematics_np, quat_to_mat, scale_model_per_body
    from uhc_amd.model.shapes import box_triangles
    from uhc_amd.sim import make_ctrl
    from uhc_amd.smpllib.smpl_robot import robot_variant
    n = 4
    rng = np.random.default_rng({"generated": 101, "shape": 102, "ball": 103, "ball_objects": 104}[name])
    qh = np.tile(standing["qpos"], (n, 1))
    qh[:, 7:] += rng.normal(scale=0.02, size=(n, 69))
    vh = rng.normal(scale=0.05, size=(n, 75))
    if name in ("generated", "shape"):
        base = model
        if name == "shape":  # per-body length scales ~ U(0.85, 1.15); the root rides at the height that puts the lowest hull vertex where the asset has it
            srng = np.random.default_rng(7)
            base = scale_model_per_body(model, np.r_[1.0, srng.uniform(0.85, 1.15, size=model.nbody - 1)])

            def lowest(m):
                xp, xq, _, _ = kinematics_np(m, standing["qpos"])
                return min((m.mesh_vert[m.geom_vertadr[g]:m.geom_vertadr[g] + m.geom_vertnum[g]] @ quat_to_mat(xq[m.geom_bodyid[g]]).T + xp[m.geom_bodyid[g]])[:, 2].min()
                           for g in range(m.ngeom) if m.geom_type[g] == 7)
            qh[:, 2] += lowest(model) - lowest(base)
        m = dataclasses.replace(robot_variant(base, {"mesh": True, "model": "smpl"}), solver=1)
        return m, make_ctrl(m), qh, vh, 0.05, qh[:, 7:].copy()
    ball = robot_variant(model, {"mesh": True, "model": "smpl", "ball": True})  # copycat_ball_1.yml's robot block
    ctrl = make_ctrl(model, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
    hb = ball_variant(model)
    if name == "ball_objects":
        ang = rng.uniform(0, 2 * np.pi, size=4)
        # (each box turned by a random 0.1 rad: axis-aligned boxes meet floor and each other in EXACT ties -- four bottom vertices at one height, face on
        #  face -- and which vertex a support query or MPR's portal picks then hangs on the last bit of a dot product: the oracle itself answers a
        #  1e-15 rotation of such a box with a 5e-5 change after one control step (round 5, step 40 of this very scene); real clips are never aligned)
        from scipy.spatial.transform import Rotation as sRot
        quats = sRot.from_rotvec(rng.normal(scale=0.1, size=(4, 3))).as_quat()[:, [3, 0, 1, 2]]
        poses = np.stack([np.r_[-0.15 + 0.75 * np.cos(a), -0.05 + 0.75 * np.sin(a), 0.18 + 0.35 * k, quats[k]] for k, a in enumerate(ang)])
        ball = add_free_bodies(ball, [box_triangles(0.15, 0.15, 0.15)] * 4, poses, density=5.0 / 0.027, friction=1.0, condim=1)
    ball = dataclasses.replace(ball, solver=1)
    q = np.tile(ball.qpos0, (n, 1))
    for e in range(n):
        q[e, :99] = hinge_to_ball_qpos(model, hb, qh[e])
    v = np.zeros((n, ball.nv))
    v[:, :75] = vh
    return ball, ctrl, q, v, 0.003, np.zeros((n, 69))  # (x a_scale x 100: torques of a few N m; the hands weigh 0.4 kg)


def _run(name, model, standing, mode, handover, steps=N_STEPS, act_scale=None, seed=7, resync=False, restart_failed=False):
    import torch
    from oracle.physics import OracleSim
    from uhc_amd import sim as S
    m, ctrl, q0, v0, a_sc, tb = _class(name, model, standing)
    a_sc = act_scale or a_sc
    n = q0.shape[0]
    old = os.environ.get("UHC_FORCE_GENERAL")
    os.environ["UHC_FORCE_GENERAL"] = "1" if mode == "general" else "0"
    try:
        b = S.SimBatch(m, ctrl, n)
    finally:
        if old is None:
            os.environ.pop("UHC_FORCE_GENERAL", None)
        else:
            os.environ["UHC_FORCE_GENERAL"] = old
    if mode == "sticky":
        b.set_kernel_path(2)
    b.set_state(torch.from_numpy(q0), torch.from_numpy(v0))
    b.sync()
    os_ = [OracleSim(m, ctrl) for _ in range(n)]
    for e in range(n):
        os_[e].set_state(q0[e], v0[e])
    tbd = torch.from_numpy(tb).cuda()
    rng = np.random.default_rng(seed)
    err = np.zeros((steps, n))
    ties = []  # resync runs: steps at which the contact model itself is discontinuous (see below)
    failed = []  # restart_failed runs: (env, step) of every bad-value flag, device and oracle
    where = np.zeros((steps, n), dtype=int)  # the qpos coordinate that carries the step's largest position error
    info = []
    for t in range(steps):
        act = rng.normal(scale=a_sc, size=(n, ctrl.action_dim))
        prev = (b.field(S.F_QPOS).cpu().numpy().copy(), b.field(S.F_QVEL).cpu().numpy().copy()) if resync else None
        b.simulate(torch.from_numpy(act).cuda(), tbd)
        b.sync()
        gq, gv = b.field(S.F_QPOS).cpu().numpy(), b.field(S.F_QVEL).cpu().numpy()
        redo, ncon, nefc = (b.field(f).cpu().numpy() for f in (S.F_REDO, S.F_NCON, S.F_NEFC))
        fail = b.field(S.F_FAIL).cpu().numpy()
        for e in range(n):
            if resync:  # one control step from the device's own state of a step ago (torque-driven classes: no controller state to carry over)
                os_[e].set_state(prev[0][e], prev[1][e])
            os_[e].do_simulation(act[e], tb[e], redo=redo[e] if handover else 0)
            dq = np.abs(gq[e] - os_[e].get("qpos"))
            dv = np.abs(gv[e] - os_[e].get("qvel"))
            if restart_failed:  # saturated random torques spin limbs up to 1e3-1e5 rad/s before the bad-value flag ends the episode: the error is taken
                dv = dv / np.maximum(1.0, np.abs(os_[e].get("qvel")))  # relative to the coordinate's size there (absolute wherever |qvel| <= 1)
            err[t, e] = max(dq.max(), dv.max())
            where[t, e] = int(dq.argmax())
            if restart_failed and (fail[e] or os_[e].geti("fail")):
                # a simulation that blew up (mj_checkPos / checkVel / checkAcc -> `fail`, uhc/envs/humanoid_im.py:1207-1211): both sides must say so in the
                # same control step; the env is put back to its start state below, as the env layer's reset would
                failed.append(dict(env=e, step=t, device=int(fail[e]), oracle=int(os_[e].geti("fail"))))
                err[t, e] = 0.0
                continue
            if resync and err[t, e] > 1e-7:
                # one control step from the SAME state, and still apart: either a defect, or a step at which the model itself is discontinuous -- a
                # support query or MPR's portal choosing between vertices that tie to the last bit (a box flat on the floor, face on face).  The
                # checker decides which: the oracle against itself, started 1e-15 away from that state.  If ITS answer moves by more than 1e-8 in
                # that one step, the step is a discontinuity of the (MuJoCo-restated) contact model and rounding picks the branch -- and the device must
                # then have landed ON one of the oracle's own branches (or, with several ties at once, inside their spread): err becomes the distance
                # to the nearest branch found, which the tests bound like every other step's error.
                sens, near = 0.0, float(err[t, e])
                pert = [s_ * 1e-15 * np.cos(k_ * np.arange(prev[0][e].shape[0] - 7)) for k_ in (1, 2, 3) for s_ in (1.0, -1.0)]
                for dp in pert:
                    pq = prev[0][e].copy()
                    pq[7:] += dp
                    twin = OracleSim(m, ctrl)
                    twin.set_state(pq, prev[1][e])
                    twin.do_simulation(act[e], tb[e])
                    sens = max(sens, np.abs(twin.get("qpos") - os_[e].get("qpos")).max(), np.abs(twin.get("qvel") - os_[e].get("qvel")).max())
                    near = min(near, max(np.abs(gq[e] - twin.get("qpos")).max(), np.abs(gv[e] - twin.get("qvel")).max()))
                ties.append(dict(env=e, step=t, err=float(err[t, e]), oracle_self_sensitivity=float(sens), err_to_nearest_oracle_branch=float(near), coordinate=int(where[t, e]),
                                 rows=int(nefc[e])))
                if sens > 1e-8:
                    # a tie the device resolved like one of the oracle's branches counts as that branch's error; with several ties at once (a box on four
                    # vertices, two pairs of hulls) the branches found need not include the device's, and the step counts as accounted for when the device is no
                    # further from the oracle than twice the oracle's own spread -- every tie is listed in rep["ties"] and bounded again by the tests
                    err[t, e] = near if near < 1e-7 else (0.0 if err[t, e] <= 2.0 * sens else float(err[t, e]))
        if restart_failed and fail.any():
            ids = np.nonzero(fail)[0]
            b.set_state(torch.from_numpy(q0[ids]), torch.from_numpy(v0[ids]), torch.from_numpy(ids.astype(np.int32)))
            b.sync()
            for e in ids:
                os_[e].set_state(q0[e], v0[e])
        info.append(dict(redo=redo.copy(), ncon=ncon.copy(), nefc=nefc.copy(), fail=fail.copy(),
                         o_ncon=np.array([o.geti("ncon") for o in os_]), o_nefc=np.array([o.geti("nefc") for o in os_]),
                         o_fail=np.array([o.geti("fail") for o in os_])))
    # ---- the report: per env, when (if ever) it leaves the tolerance, and what the step before / at the exit looked like
    rep = dict(workload=name, mode=mode, handover=bool(handover), resync_every_step=bool(resync), steps=steps, action_scale=a_sc, worst=float(np.nanmax(err)),
               env_steps_primal=int(sum(((i["redo"] & (1 << 30)) != 0).sum() for i in info)), env_steps_primal_cap=int(sum(((i["redo"] & (1 << 29)) != 0).sum() for i in info)),
               worst_first_50=float(np.nanmax(err[:50])), nefc_max=int(max(i["nefc"].max() for i in info)), ncon_max=int(max(i["ncon"].max() for i in info)),
               env_steps_general_or_large=int(sum((i["redo"] & 1).sum() for i in info)), env_steps_large=int(sum(((i["redo"] & 0x40) != 0).sum() for i in info)),
               env_steps_swept=int(sum(((i["redo"] & 2) != 0).sum() for i in info)), env_steps_windowed=int(sum(((i["redo"] & 8) != 0).sum() for i in info)),
               env_steps_rows_dropped=int(sum(((i["redo"] & 0x80) != 0).sum() for i in info)), ties=ties, failed=failed,
               env_steps_primal_by_rows=sorted(int(x) for i in info for x in i["nefc"][(i["redo"] & (1 << 30)) != 0])[-16:], leaves=[])
    for e in range(n):
        bad = np.nonzero(~(err[:, e] < TOL))[0]
        if bad.size == 0:
            continue
        t = int(bad[0])
        before = [i for i in info[:t + 1]]
        why = []
        if any((i["redo"][e] & 0x80) for i in before):
            why.append("rows dropped beyond the last tier's capacity in an earlier step")
        if any((i["redo"][e] & 2) for i in before):
            why.append("an exact solve fell back to sweeps" + ("" if handover else " (nothing handed over: the oracle stayed exact)"))
        flips = [k for k, i in enumerate(before) if i["ncon"][e] != i["o_ncon"][e] or i["nefc"][e] != i["o_nefc"][e]]
        if flips:
            why.append(f"contact / row sets differ from step {flips[0]} on (device {int(info[flips[0]]['ncon'][e])} contacts / {int(info[flips[0]]['nefc'][e])} rows, "
                       f"oracle {int(info[flips[0]]['o_ncon'][e])} / {int(info[flips[0]]['o_nefc'][e])}); error one step earlier {err[max(flips[0] - 1, 0), e]:.1e}")
        if any(i["fail"][e] or i["o_fail"][e] for i in before):
            why.append("bad-value flag raised")
        if not why:
            why.append("same contact sets, same solver: rounding amplified by the dynamics")
        # how the error grew: the first step above each decade, and the coordinate that carried it when it left (humanoid: < qpos_lim; beyond: an object)
        decades = {f"1e{k}": int(np.nonzero(err[:, e] > 10.0 ** k)[0][0]) for k in (-12, -10, -8, -6, -4) if (err[:, e] > 10.0 ** k).any()}
        nqh = 99 if name.startswith("ball") else 76
        rep["leaves"].append(dict(env=e, step=t, err=float(err[t, e]), err_10_steps_before=float(err[max(t - 10, 0), e]), why=why, first_step_above=decades,
                                  worst_coordinate=int(where[t, e]), worst_is=("humanoid root" if where[t, e] < 7 else "humanoid joint" if where[t, e] < nqh else f"object {(where[t, e] - nqh) // 7}"),
                                  rows_at_exit=[int(info[t]["nefc"][e]), int(info[t]["o_nefc"][e])], redo_at_exit=hex(int(info[t]["redo"][e]))))
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity200.jsonl"), "a") as f:
        f.write(json.dumps(rep) + "\n")
    print(json.dumps(rep))
    b.close()
    return rep


@pytest.mark.parametrize("handover", [True, False], ids=["handover", "oracle_decides"])
@pytest.mark.parametrize("mode", ["fast", "general", "sticky"])
@pytest.mark.parametrize("name", ["generated", "shape"])
def test_200_steps_within_1e_4(model, standing, name, mode, handover):
    rep = _run(name, model, standing, mode, handover)
    assert rep["env_steps_rows_dropped"] == 0 and rep["env_steps_primal_cap"] == 0, rep
    if not handover:
        assert rep["env_steps_swept"] == 0, rep  # the device never needed the sweeps: nothing to hand over
    assert not rep["leaves"] and rep["worst"] < TOL, rep


@pytest.mark.parametrize("mode", ["fast", "general", "sticky"])
@pytest.mark.parametrize("name", ["ball", "ball_objects"])
def test_200_steps_of_the_ball_joint_classes(model, standing, name, mode):
    """(a) free run, nothing handed over: within 1e-7 for the first 25 control steps; where it leaves 1e-4 is reported (the oracle against a copy of
    itself 1e-14 away leaves at steps 57-72: profiles/r05_chaos_probe.txt).  (b) the same 200 steps, the oracle re-started from the device's state
    before every step: every control step within 1e-9 -- through standing, falling, lying among the boxes, whichever tier computed it.
    Neither run may drop a row or sweep: every solve is exact (working sets, or Newton on the primal in tier 4)."""
    free = _run(name, model, standing, mode, False)
    assert free["env_steps_rows_dropped"] == 0 and free["env_steps_swept"] == 0 and free["env_steps_primal_cap"] == 0, free
    first = min([l["step"] for l in free["leaves"]] + [N_STEPS])
    early = min([l["first_step_above"].get("1e-8", N_STEPS) for l in free["leaves"]] + [N_STEPS])
    assert first >= 35 and early >= 20, (first, early, free)  # (1e-8 is reached at steps 24-35, 1e-4 at 41-73 on every box so far)
    step = _run(name, model, standing, mode, False, resync=True)
    assert step["env_steps_rows_dropped"] == 0 and step["env_steps_swept"] == 0 and step["env_steps_primal_cap"] == 0, step
    # every control step within 1e-7 (qvel carries the step's largest error: 1e-9 in qpos), except steps the oracle itself marks as discontinuities of the
    # contact model (rep["ties"]: its own answer moves by > 1e-8 under a 1e-15 perturbation of the start state) -- a handful per 800 env-steps
    assert step["worst"] < 1e-7 and not step["leaves"], step
    # (round 6: a tie is no longer written off -- its error is the distance to the nearest branch the oracle itself takes under 1e-15 perturbations, bounded
    #  by `worst` above like every other step; what is left to bound here is how many there are, and that each really is a discontinuity of the checker)
    assert all(t["oracle_self_sensitivity"] > 1e-8 and t["err_to_nearest_oracle_branch"] <= max(1e-7, 2 * t["oracle_self_sensitivity"]) for t in step["ties"]) and len(step["ties"]) <= 40, step["ties"]


@pytest.mark.parametrize("name", ["ball", "ball_objects"])
def test_ball_joint_rollout_at_policy_scale_torques_reports_where_it_leaves(model, standing, name):
    """The ball-joint configs drive torques directly (copycat_ball_1.yml: action_type torque): an init-policy action of sigma 0.1 is
    0.1 x a_scale x 100 = thousands of N m before the clip at 4 x torque_lim -- every motor saturated with a random sign, 30 times a
    second, on a humanoid without joint limits.  That is what bench.py's `ball_rollout` / `configs4` probes run: the humanoid folds into itself
    (400-470 rows: tier 4), some envs blow up (bad-value flag) within 15 steps.  The free run is reported; asserted: no rows dropped, no sweeps,
    and no env leaves 1e-4 within the first 10 control steps."""
    rep = _run(name, model, standing, "sticky", False, steps=60, act_scale=0.1, seed=9)
    assert rep["env_steps_rows_dropped"] == 0 and rep["env_steps_swept"] == 0, rep
    first = min([l["step"] for l in rep["leaves"]] + [60])
    assert first >= 10, rep


@pytest.mark.parametrize("name", ["ball", "ball_objects"])
def test_ball_joint_rollout_at_policy_scale_torques_step_by_step(model, standing, name):
    """VERDICT r5 "next" 4: parity evidence at the action scale bench.py's `ball_rollout` / `configs4` probes run (sigma 0.1: every motor saturated,
    `torque = ctrl * a_scale * 100` clipped, uhc/envs/humanoid_im.py:1158-1160), not only for the first ten steps of a free run.  60 control steps through the
    sticky queues, the oracle re-started from the device's state before every step: every control step within 1e-7, hundreds of rows through tier 4
    (Newton on the primal, the four-wave consumers when the host has started them); an env whose simulation blows up raises `fail` on BOTH sides in the
    same step, is left out of that step's error and restarted (as the env layer's reset does).  No rows dropped, no sweeps, Newton never at its cap."""
    rep = _run(name, model, standing, "sticky", False, steps=60, act_scale=0.1, seed=9, resync=True, restart_failed=True)
    assert rep["env_steps_rows_dropped"] == 0 and rep["env_steps_swept"] == 0 and rep["env_steps_primal_cap"] == 0, rep
    assert all(f["device"] == 1 and f["oracle"] == 1 for f in rep["failed"]), rep["failed"]
    assert rep["worst"] < 1e-7 and not rep["leaves"], rep
    assert all(t["oracle_self_sensitivity"] > 1e-8 and t["err_to_nearest_oracle_branch"] <= max(1e-7, 2 * t["oracle_self_sensitivity"]) for t in rep["ties"]) and len(rep["ties"]) <= 12, rep["ties"]
    # the point of the test: tier 4 did the work, on hundreds of rows
    # (saturated random torques end an episode within 10-20 control steps: tier 4 sees the last few of each -- the humanoid folded into itself, 400+ rows)
    assert rep["env_steps_primal"] >= 3 and max(rep["env_steps_primal_by_rows"] + [0]) >= 300 and len(rep["failed"]) >= 3, (rep["env_steps_primal"], rep["nefc_max"], rep["failed"])
