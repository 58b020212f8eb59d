"""Where one control step's time goes ACROSS the tiers (sticky tiers, self-colliding rollout): per env, when each tier took it up and
let it go, and at which substep it was handed on.  Needs UHC_DEBUG bit 4 (set here): the product kernel then stamps a 100 MHz wall
clock into the first words of the env's stage-profile record (uhc_physics_impl.h, TRACE).

  python tools/tier_trace.py [out.txt] [flags of bench.py]           the self-colliding rollout (bench line `self_collision`)
  TRACE_BALL=1 python tools/tier_trace.py [out.txt]                   the ball-joint humanoid's rollout (`ball_rollout`)
  python tools/tier_trace.py [out.txt] --workload ball_objects        the physics-only scene with objects (`ball_objects`)
"""
import os
import sys

os.environ["UHC_DEBUG"] = str(int(os.environ.get("UHC_DEBUG", "0")) | 16)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import bench
from uhc_amd import sim as S


def analyze(prof, n, step, lines):
    t = prof[:, :8].cpu().numpy().astype(np.float64)
    stamps = t[:, :6]
    t0 = stamps[stamps > 0].min()
    ms = np.where(stamps > 0, (stamps - t0) * 1e-5, np.nan)  # 100 MHz -> ms
    end = np.nanmax(ms, axis=1)
    last_tier = np.where(~np.isnan(ms[:, 5]), 3, np.where(~np.isnan(ms[:, 3]), 2, 1))
    lines.append(f"## step {step}: {n} envs, step ends at {np.nanmax(end):.2f} ms; finished in tier 1 / 2 / 3: "
                 f"{(last_tier == 1).sum()} / {(last_tier == 2).sum()} / {(last_tier == 3).sum()}")
    for tier in (1, 2, 3):
        s, e = ms[:, 2 * tier - 2], ms[:, 2 * tier - 1]
        m = ~np.isnan(s)
        if not m.any():
            continue
        q = lambda x: " ".join(f"{v:6.2f}" for v in np.nanpercentile(x, [0, 25, 50, 75, 90, 99, 100]))
        lines.append(f"tier {tier}: {m.sum():4d} envs | start  (min 25% 50% 75% 90% 99% max) {q(s[m])} | end {q(e[m])} | duration {q((e - s)[m])}")
    h1 = ~np.isnan(ms[:, 2]) & ~np.isnan(ms[:, 0])   # handed on by tier 1 this step
    if h1.any():
        wait = ms[h1, 2] - ms[h1, 1]
        lines.append(f"handed on by tier 1: {h1.sum()} envs; at substep (mean) {t[h1, 6].mean():.1f}; hand-on time {np.percentile(ms[h1, 1], [0, 50, 100]).round(2).tolist()} ms;"
                     f" wait for a consumer {np.percentile(wait, [0, 50, 100]).round(2).tolist()} ms")
    h2 = ~np.isnan(ms[:, 4]) & ~np.isnan(ms[:, 2])
    if h2.any():
        wait = ms[h2, 4] - ms[h2, 3]
        lines.append(f"handed on by tier 2: {h2.sum()} envs; at substep (mean) {t[h2, 7].mean():.1f}; hand-on time {np.percentile(ms[h2, 3], [0, 50, 100]).round(2).tolist()} ms;"
                     f" wait for a consumer {np.percentile(wait, [0, 50, 100]).round(2).tolist()} ms")
    t4 = prof[:, 21:24].cpu().numpy().astype(np.float64)  # tier 4's own stamps (taken up, let go) and the substep at which the large tier handed the env on
    m4 = t4[:, 0] > 0
    if m4.any():
        s4, e4 = (t4[m4, 0] - t0) * 1e-5, (t4[m4, 1] - t0) * 1e-5
        lines.append(f"tier 4: {int(m4.sum()):4d} envs | start {np.round(np.sort(s4), 2).tolist()} | end {np.round(np.sort(e4), 2).tolist()} | duration {np.round(np.sort(e4 - s4), 2).tolist()} ms"
                     f" | handed on by the large tier at substep {t4[m4, 2].astype(int).tolist()}")
        end = np.where(m4, np.fmax(end, (t4[:, 1] - t0) * 1e-5), end)
        lines.append(f"        (with tier 4's envs the step ends at {np.nanmax(end):.2f} ms)")
    full = prof[:, :16].cpu().numpy().astype(np.float64)
    for tier, o in ((2, 8), (3, 12)):
        c = full[:, o:o + 4]
        m = c[:, 0] > 0
        if m.any():
            rel = lambda x: (x - t0) * 1e-5
            first = np.where(c[m, 1] > 0, rel(c[m, 1]), np.nan)
            lines.append(f"tier {tier} consumers: {m.sum()} workgroups; entry {np.percentile(rel(c[m, 0]), [0, 50, 100]).round(2).tolist()} ms; first env claimed "
                         f"{np.nanpercentile(first, [0, 50, 100]).round(2).tolist() if (~np.isnan(first)).any() else 'never'} ms ({int(np.isnan(first).sum())} never); "
                         f"exit {np.percentile(rel(c[m, 2]), [0, 50, 100]).round(2).tolist()} ms; envs per workgroup {np.percentile(c[m, 3], [0, 50, 100]).tolist()}")
    g = prof[n - 1, 16:21].cpu().numpy().astype(np.float64)
    if g[0] > 0:
        lines.append(f"gate: from {(g[0] - t0) * 1e-5:.3f} to {(g[1] - t0) * 1e-5:.3f} ms; consumers reported in at its start / end {int(g[2])} / {int(g[3])} of {int(g[4])}")
    worst = np.argsort(-end)[:6]
    for w in worst:
        lines.append(f"  env {w:4d} ends {end[w]:6.2f} ms: " + " ".join(
            f"T{k + 1}[{ms[w, 2 * k]:.2f}..{ms[w, 2 * k + 1]:.2f}]" for k in range(3) if not np.isnan(ms[w, 2 * k])) +
            f" hand-on substeps {int(t[w, 6])} {int(t[w, 7])}")


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else None
    sys.argv = ["bench.py"] + [a for a in sys.argv[1:] if a != out_path]
    args = bench.parse()
    torch.cuda.set_device(0)
    torch.set_default_dtype(torch.float64)
    lines = []
    if args.workload == "ball_objects":  # physics only: the scene of bench.py --workload ball_objects, traced on a few of its timed steps
        which = {3, 17, 28, 29}
        def pre(sim, i):
            if i in which:
                sim.field(S.F_STAGE_PROF).zero_()
                torch.cuda.synchronize()
        def post(sim, i):
            if i in which:
                torch.cuda.synchronize()
                analyze(sim.field(S.F_STAGE_PROF), sim.n_env, i, lines)
        args.hooks = (pre, post)
        args.steps, args.warmup = 30, 20
        bench.bench_ball_objects(args)
    else:
        if os.environ.get("TRACE_PROBE"):  # any probe of bench.py by its key (configs4, ball_rollout, floor_only, shapes)
            pr = dict(bench.PROBES[os.environ["TRACE_PROBE"]])
            pr.pop("name")
            agent = bench.build_agent(args, 0, 0, torch.float64, **pr)
        elif os.environ.get("TRACE_BALL") == "1":  # the ball-joint humanoid of the bench line `ball_rollout`
            agent = bench.build_agent(args, 0, 0, torch.float64, robot_cfg={"mesh": True, "model": "smpl", "ball": True},
                                      cfg_over=dict(action_type="torque", residual_force=False, meta_pd=False, meta_pd_joint=False, reward_id="world_rfc_implicit_quat",
                                                    obs_v=2, tq_mul=4, env_init_noise=0.0))
        else:
            agent = bench.build_agent(args, 0, 0, torch.float64, robot_cfg={"mesh": True, "model": "smpl"})
        agent.per_epoch_update(0)
        env = agent.env
        warm = int(os.environ.get("TRACE_WARM", "12"))  # (12: inside the transient after the restart of all envs; 50: steady state)
        agent.rollout_begin(warm + 28)
        for _ in range(warm):
            agent.rollout_step()
        torch.cuda.synchronize()
        prof = env.sim.field(S.F_STAGE_PROF)
        why_hist = np.zeros((2, 6), dtype=np.int64)  # per tier: contacts, rows, body-body slots, row storage, candidate list, (any)
        sub_hist = np.zeros(16, dtype=np.int64)
        for step in range(24):
            if step < 4:
                prof.zero_()
            torch.cuda.synchronize()
            agent.rollout_step()
            torch.cuda.synchronize()
            if step < 4:
                analyze(prof, env.n_env, step, lines)
            w = env.sim.field(S.F_HANDON_WHY).cpu().numpy()
            for tier in range(2):
                b = (w >> (8 * tier)) & 0xff
                for k in range(5):
                    why_hist[tier, k] += int(((b >> k) & 1).sum())
                why_hist[tier, 5] += int((b != 0).sum())
            s = (w >> 16) & 0xff
            for k in np.unique(s[(w & 0xffff) != 0]):
                sub_hist[min(int(k) if k < 128 else 0, 15)] += int(((s == k) & ((w & 0xffff) != 0)).sum())
        lines.append(f"## hand-ons over 24 steps x {env.n_env} envs (UHC_F_HANDON_WHY; an env can name several reasons): [contacts, rows, body-body row slots, packed row storage, MPR candidate list, envs]")
        lines.append(f"fast tier -> general: {why_hist[0].tolist()};  general tier -> large: {why_hist[1].tolist()};  substep of the last hand-on (0 = the reset's forward pass / first substep): {sub_hist.tolist()}")
    txt = "\n".join(lines)
    print(txt)
    if out_path:
        open(out_path, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
