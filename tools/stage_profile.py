"""Per-stage shader-cycle breakdown of the fused step kernel.  Builds libuhc_amd_prof.so with
-DUHC_STAGE_PROF and runs the bench workload for a few steps.  Usage (GPU box):
    python tools/stage_profile.py [n_env] [steps]"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "uhc_amd", "csrc")
PROF_LIB = os.path.join(CSRC, "libuhc_amd_prof.so")
NAMES = ["pd+rfc", "kinematics", "com_pos", "crb", "factor", "com_vel", "rne", "smooth", "collision", "rows", "A-build",
         "pgs-sweeps", "z+rest/pgs-general", "qacc-solve", "euler", "store",
         "pd: M->LD + gains", "pd: factor", "pd: solve", "kin: pass 1 (local poses)", "kin: pass 2 (levels)", "crb: subtree sums", "crb: I*cdof",
         "rne: levels", "pd: M -> LD", "as: W load (+loop tail)", "as: elimination", "as: back substitution", "as: y = A f + b", "as: pre-sweeps", "ws: islands + compaction + row load", "ws: y on the other rows",
         "col: plane-mesh pairs", "col: convex pairs sphere cull", "col: vertex staging", "col: MPR",
         "t4: jar = Yhat u + b", "t4: active set + rank-one updates", "t4: chain rows -> gradient (+ Hessian)", "t4: dense rows -> gradient + Hessian"]
# (tier 4 reuses the slots of the dual solver's stages: 26 = Cholesky, 27 = substitutions, 28 = p = Yhat dir + line search, 30 = start point, 31 = gradient norm / rest)


def build():
    import __graft_entry__ as g
    g.compile_lib(lib=PROF_LIB, extra_flags=["-DUHC_STAGE_PROF"], obj_dir=os.path.join(CSRC, "build_prof"))


if __name__ == "__main__":
    if "--build-only" in sys.argv:
        build()
        sys.exit(0)
    if "UHC_LIB" not in os.environ:
        if not os.path.exists(PROF_LIB):
            build()
        os.environ["UHC_LIB"] = PROF_LIB
    import numpy as np
    import torch
    from uhc_amd import sim as S
    n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    import dataclasses
    model = dataclasses.replace(S.load_asset_model(), solver=int(os.environ.get("SOLVER", "0")), iterations=int(os.environ.get("CAP", "100")))
    z = np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz"))
    rng = np.random.default_rng(1)
    scene = os.environ.get("MODEL", "copycat")  # copycat | selfcol (body-body collisions on) | ball_objects (bench.py --workload ball_objects)
    if scene == "ball_objects":
        from uhc_amd.model.shapes import box_triangles
        from uhc_amd.model.mjcf import add_free_bodies, ball_variant, hinge_to_ball_qpos, self_collision_variant
        base = model
        K = int(os.environ.get("OBJECTS", "4"))
        ang = np.random.default_rng(11).uniform(0, 2 * np.pi, size=K)
        poses = np.stack([np.r_[-0.15 + 0.75 * np.cos(a), -0.05 + 0.75 * np.sin(a), 0.3 + 0.45 * k, 1, 0, 0, 0] for k, a in enumerate(ang)]) if K else np.zeros((0, 7))
        hb = ball_variant(base, damping=5.0)
        model = self_collision_variant(hb)
        if K:
            model = add_free_bodies(model, [box_triangles(0.15, 0.15, 0.15)] * K, poses, density=5.0 / 0.027)
        model = dataclasses.replace(model, solver=base.solver, iterations=base.iterations)
        ctrl = S.make_ctrl(base, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
        qpos = np.tile(model.qpos0, (n_env, 1))
        for e in range(n_env):
            qh = z["qpos"].copy()
            qh[7:] += rng.normal(scale=0.1, size=69)
            qpos[e, :99] = hinge_to_ball_qpos(base, hb, qh)
        qvel = np.zeros((n_env, model.nv))
        qvel[:, :75] = rng.normal(scale=0.2, size=(n_env, 75))
        actions = 0.1 * rng.normal(size=(8, n_env, ctrl.action_dim))
    else:
        if scene == "selfcol":
            from uhc_amd.model.mjcf import self_collision_variant
            model = dataclasses.replace(self_collision_variant(model), solver=model.solver, iterations=model.iterations)
        ctrl = S.make_ctrl(model)
        qpos = np.tile(z["qpos"], (n_env, 1))
        qpos[:, 7:] += rng.normal(scale=0.05, size=(n_env, model.nu))
        qvel = rng.normal(scale=0.1, size=(n_env, model.nv))
        actions = rng.normal(scale=np.exp(-2.3), size=(8, n_env, ctrl.action_dim))
    b = S.SimBatch(model, ctrl, n_env)
    b.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel))
    b.sync()
    b.field(S.F_STAGE_PROF).zero_()
    a = torch.from_numpy(actions).cuda()
    tb = torch.from_numpy(np.ascontiguousarray(qpos[:, 7:7 + 69])).cuda()
    for t in range(steps):
        b.simulate(a[t % 8], tb)
    b.sync()
    p = b.field(S.F_STAGE_PROF).cpu().numpy().astype(np.float64) / steps
    tot = p.sum(1)
    print(f"scene={scene} general_share={(b.field(S.F_REDO) != 0).float().mean().item():.2f} ", end="")
    print(f"n_env={n_env} steps={steps}: mean cycles per env-step = {tot.mean():.0f} (max {tot.max():.0f}); nefc mean {b.field(S.F_NEFC).float().mean().item():.1f} iters mean {b.field(S.F_SOLVER_ITER).float().mean().item():.1f}")
    for k, nm in enumerate(NAMES):
        if p[:, k].mean() > 0:
            print(f"  {nm:26s} {p[:, k].mean():12.0f} cycles/env-step  {100 * p[:, k].mean() / tot.mean():5.1f}%")
    # a launch lasts as long as its slowest env: the same breakdown for the slowest 1 % of the envs (top-level stages only)
    top = p[:, :16].sum(1)
    slow = np.argsort(top)[-max(n_env // 100, 1):]
    print(f"slowest 1% of the envs: {top[slow].mean():.0f} cycles/env-step in the top-level stages (all envs: {top.mean():.0f}); nefc of their last step: {b.field(S.F_NEFC)[torch.from_numpy(slow).cuda()].tolist()}")
    for k, nm in enumerate(NAMES):
        if p[:, k].mean() > 0:
            print(f"  {nm:26s} {p[slow, k].mean():12.0f} cycles/env-step  x{p[slow, k].mean() / max(p[:, k].mean(), 1):.2f} of the mean")
