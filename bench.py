"""bench.py -- hot-path throughput on MI355X (contract: task statement; design notes: DESIGN.md section 6).

A "step" is one control step of EVERY environment of the batch, exactly as a training rollout performs it:
observation filter -> policy MLP forward (sampling) -> PD-target gather -> fused physics kernel (15 substeps of
stable-PD, residual force, forward dynamics, contact solve, Euler) -> termination, imitation reward and next
observation -> rollout-buffer writes -> reset of finished episodes (new clip window, set_state, forward).
Workload = BASELINE.json configs[1]: copycat config, 1024 batched envs per GPU, synthetic clips, on the model class the reference's
env runs -- what Robot(cfg.robot_cfg) generates: body-body collisions on, Chest / shoulder excludes, rel_joint_lm joint ranges
(uhc/smpllib/smpl_parser.py:327-328, smpl_robot.py:1087-1110, 1177-1198).  `--floor-only` runs the shipped static asset as it is
(floor contacts only: rounds 1-3's headline; kept as the `floor_only` sub-line).
value = env-steps/s summed over all ranks (weak scaling: envs shard across ranks, no data-path collective).
After the timed region one full PPO update (GAE + 10 full-batch epochs, gradients all-reduced over RCCL when
n_gpus > 1) over the collected samples is timed and reported as ppo_samples_per_s.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("OMP_PROC_BIND", "close")  # cpu_baseline leg: pin the oracle's OpenMP threads (set before any OpenMP runtime loads)
os.environ.setdefault("OMP_PLACES", "cores")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic HBM bytes of one env-step of the fused physics kernel:
# read qpos, qvel, warm start, action, expert target; write qpos, qvel, qacc, body pos / quat / com
ALGO_BYTES_PER_ENV_STEP = 8 * (76 + 75 + 75 + 105 + 69 + 76 + 75 + 75 + 75 + 100 + 75)
HBM_PEAK_GBS = 8000.0


FAST_KERNEL = {False: "void uhc_step_kernel<0, 1, false>", True: "void uhc_step_kernel<0, 1, true>"}  # floor-only | with body-body contacts compiled in


def pmc_traffic(kernel, tag=""):
    """HBM-side bytes per launch of a step kernel from the committed rocprofv3 PMC passes of the same workload
    (profiles/*_pmc_{FETCH,WRITE}_SIZE<tag>.txt, written by tools/profile_on_gpu.sh; the counters are reported in KB).  PMC
    collection needs rocprofv3 around the process, so the live run quotes the latest committed pass (null if none)."""
    import glob
    import re
    tot, src = 0.0, []
    for kind in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_{kind}{tag}.txt")),
                       key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])  # r01_v12 after r01_v9
        if not files:
            return None, None
        for line in open(files[-1]):
            if line.startswith(kernel) and f"| {kind} |" in line:
                tot += float(line.split("|")[2]) * 1024.0
                src.append(os.path.basename(files[-1]))
                break
        else:
            return None, None
    return tot, "+".join(src)


def _latest(pattern):
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    return files[-1] if files else None


def contact_solve_share(tag=""):
    """Share of the fused kernel's cycles spent in the contact solve proper (active-set factorisations + pre-sweeps, or the PGS sweeps),
    from the latest committed stage profile (tools/stage_profile.py, an instrumented build: cannot run inside the timed region)."""
    f = _latest(f"*_stage_profile{tag}.txt")
    if not f:
        return None, None
    share = 0.0
    for line in open(f):
        name = line.strip().split("  ")[0]
        if line.startswith("slowest"):
            break  # (the per-stage table of the slowest envs follows: ratios, not shares)
        if (name.startswith("as:") or name.startswith("pgs-sweeps")) and line.rstrip().endswith("%"):
            share += float(line.strip().split()[-1].rstrip("%")) / 100.0
    return (share or None), os.path.basename(f)


def rocprof_avg_ms(kernel, tag=""):
    """Average launch duration (ms) of a step kernel in the latest committed `rocprofv3 --kernel-trace --stats` summary of the same workload
    (profiles/*_kernel_stats<tag>.txt): printed beside the HIP-event figure of this run so that the two can be compared on the line itself."""
    try:
        f = _latest(f"*_kernel_stats{tag}.txt")
        if not f:
            return None, None
        for line in open(f):
            if line.startswith(kernel) and "|" in line:
                return float(line.split("|")[3]), os.path.basename(f)
    except Exception:
        pass
    return None, None


def valu_f64_counters(kernel, tag=""):
    """float64 VALU instructions per launch of a step kernel from the latest committed PMC pass (profiles/*_pmc_VALU_F64<tag>.txt):
    flop = (ADD + MUL + TRANS + 2 FMA) wave-instructions x 64 lanes (an upper bound on useful flops: inactive lanes count too)."""
    f = _latest(f"*_pmc_VALU_F64{tag}.txt")
    if not f:
        return None
    c = {}
    for line in open(f):
        if line.startswith(kernel) and "|" in line:
            parts = [x.strip() for x in line.split("|")]
            c[parts[1]] = float(parts[2])
    need = ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64")
    if not all(k in c for k in need):
        return None
    flop = 64.0 * (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0) + 2.0 * c["SQ_INSTS_VALU_FMA_F64"])
    # active lanes: SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = lanes switched on per VALU cycle, averaged over ALL VALU instructions of the
    # kernel (the counters do not split it by type): flop x that / 64 is the estimate of the useful flops beside the every-lane-counts upper bound
    lanes = (c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]) if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU") else None
    return {"flop_per_launch": flop, "active_lanes_per_valu_cycle": lanes, "flop_per_launch_active_lanes": (flop * min(lanes, 64.0) / 64.0) if lanes else None,
            "counters": {k: v for k, v in c.items() if "F64" in k or k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU")}, "source": os.path.basename(f)}


def alu_per_env_step(name):
    """float64 flop per env-step of a whole workload -- every tier's step kernels added up -- from the committed counter pass of that
    workload (profiles/*_alu_<name>.json, written by tools/pmc_alu.py from a rocprofv3 --pmc run of the probe): what the probes' ALU
    roofline uses, because their env-steps are spread over three kernels that run side by side."""
    f = _latest(f"*_alu_{ {'configs2_per_gpu': 'headline'}.get(name, name) }.json")  # (the same workload at 4096 envs: flop per env-step of the 1024-env pass)
    if not f:
        return None
    d = json.load(open(f))
    d["source"] = os.path.basename(f)
    return d


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--envs", type=int, default=1024, help="environments per GPU")
    p.add_argument("--clips", type=int, default=64, help="synthetic clips per rank")
    p.add_argument("--ppo-dtype", default="float64", choices=["float64", "float32"])
    p.add_argument("--pgs-iterations", type=int, default=None, help="sweep cap of the contact solve (default: the config's, 300 = converged)")
    p.add_argument("--solver", type=int, default=None, choices=[0, 1], help="contact solver: 0 PGS sweeps, 1 exact active-set solve (default: the config's)")
    p.add_argument("--shapes", type=int, default=0, help="configs[3] (smpl_shape): K body shapes with per-body length scales ~ U(0.85, 1.15), default_rng(7); "
                   "every clip gets its own (clips >= K + 1 are generated), so --shapes 1023 gives 1024 distinct model blobs")
    p.add_argument("--workload", default="copycat", choices=["copycat", "ball_objects"],
                   help="copycat = configs[1] (the metric's config); ball_objects = configs[4] stand-in: ball-joint humanoid, self-collision, free boxes, torque actions (physics only)")
    p.add_argument("--objects", type=int, default=4, help="ball_objects: free boxes per env")
    p.add_argument("--general-only", action="store_true", help="ball_objects: skip the fast kernel (uhc_batch_set_kernel_path 1)")
    p.add_argument("--fixed-path", action="store_true", help="ball_objects: fast kernel then general kernel on every step (uhc_batch_set_kernel_path 0) instead of the adaptive default")
    p.add_argument("--no-pgs-probe", action="store_true", help="skip the short PGS (solver 0) kernel timings after the timed region")
    p.add_argument("--no-probes", action="store_true", help="skip the floor_only / shapes / ball_rollout / configs4 sub-lines after the timed region")
    p.add_argument("--floor-only", action="store_true", help="headline on the shipped static asset as it is (floor contacts only, +-180 degree joint ranges: "
                   "config/uhc_amd/copycat_mi355x.yml) instead of the model class the reference generates")
    p.add_argument("--probe-steps", type=int, default=60, help="timed steps per repetition of a probe")
    p.add_argument("--probe-warmup", type=int, default=40, help="untimed steps of a probe before its first repetition (outlasts the transient after the restart of all envs)")
    p.add_argument("--probe-reps", type=int, default=3, help="repetitions of a probe; the sub-line reports the median")
    p.add_argument("--only-probe", default=None, help="run one probe alone and print its sub-line (profiling: floor_only | shapes | ball_rollout | configs4)")
    p.add_argument("--preroll", type=int, default=40, help="untimed rollout steps BEFORE --warmup: all envs restart together at the head of a pass and the first one or two "
                   "episode lengths (24 steps under the random-init policy) are a transient with fewer contacts than the steady state -- the driver's "
                   "--warmup 5 alone measured +6 %% (VERDICT r4); the headline is the steady state")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-ppo", action="store_true")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU plumbing checks)")
    p.add_argument("--same-device", action="store_true", help="plumbing check: every rank uses GPU 0")
    return p.parse_args()


def cpu_baseline(agent, n_env_gpu):
    """The CPU oracle (own restatement of the MuJoCo step + PD; kind 'port') on the host cores, bounded sample of the same workload:
    physics control steps from the clips' first frames with init-policy action noise -- at ONE thread and at all host cores."""
    import ctypes as C
    from oracle.physics import OracleSim, lib
    env = agent.env
    # the threads this process may actually run on: the affinity mask, cut down to the cgroup's CPU quota where one is set (a lease of a
    # few cores on a 256-thread host reports 256 CPUs either way; 256 OpenMP threads on an 8-core quota are throttled to a crawl)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2: "max 100000" or "<quota> <period>"
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())  # cgroup v1: -1 = unlimited
            if q > 0:
                quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    affinity = cores
    if quota is not None:
        cores = max(1, min(cores, int(quota + 0.5)))
    frames = env.env._bank[0].cpu().numpy()
    starts = env.env._bank[1].cpu().numpy()
    L = lib()
    L.orc_set_threads.argtypes = [C.c_int]

    def run(n, threads, budget_s, max_steps):
        rng = np.random.default_rng(0)
        sims, tb = [], []
        for e in range(n):
            f0 = frames[starts[e % len(starts)]]
            s = OracleSim(env.model, env.ctrl)
            s.set_state(f0[0:76], frames[starts[e % len(starts)] + 1][76:151])
            sims.append(s)
            tb.append(frames[starts[e % len(starts)] + 1][7:76])
        ptrs = (C.c_void_p * n)(*[s.d for s in sims])
        tb = np.ascontiguousarray(np.stack(tb))
        L.orc_set_threads(threads)
        steps, t0 = 0, time.perf_counter()
        while True:
            a = np.ascontiguousarray(rng.normal(scale=np.exp(-2.3), size=(n, env.action_dim)))
            L.orc_batch_do_simulation(C.byref(sims[0].desc), C.byref(env.ctrl), ptrs, n, a.ctypes.data_as(C.POINTER(C.c_double)),
                                      tb.ctypes.data_as(C.POINTER(C.c_double)))
            steps += 1
            el = time.perf_counter() - t0
            if el > budget_s or steps >= max_steps:
                break
        return n * steps / el, n, steps

    v1, n1, s1 = run(8, 1, 5.0, 15)
    n = min(n_env_gpu, 4 * cores)
    vall, na, sa = run(n, cores, 10.0, 15)
    eff = vall / (cores * v1) if v1 > 0 else None
    note = "" if eff is None or eff >= 0.5 else (f"; the all-thread run reaches only {eff:.2f} of {cores} x the one-thread rate: the lease's host threads are shared / SMT siblings, "
                                                 "so `value` understates a dedicated host")
    return {"value": vall, "unit": "env-steps/s", "cores": cores, "kind": "port", "value_1_thread": v1, "scaling_efficiency": eff,
            "os_cpu_count": os.cpu_count(), "sched_affinity": affinity, "cgroup_cpu_quota": quota, "omp_proc_bind": os.environ.get("OMP_PROC_BIND"),
            "sample": f"{na} envs x {sa} physics control steps (15 substeps, PD + RFC) of the same clips on {cores} OpenMP threads (sched_getaffinity), and {n1} envs x {s1} on one thread; "
                      f"oracle/physics_oracle.c; MuJoCo itself is not installed" + note}


def cpu_ppo_baseline(agent, batch):
    """PyTorch-CPU float64 full-batch PPO epochs (value step + policy step: forward, backward, Adam) on a bounded sample of the collected
    batch -- the learner half of the reference's CPU path -- with one torch thread per physical core.  samples/s of a 10-epoch update."""
    import copy
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        phys = os.cpu_count() or 1
    n = min(4096, batch.states.shape[0])
    torch.set_num_threads(phys)
    pol, val = copy.deepcopy(agent.policy_net).cpu().double(), copy.deepcopy(agent.value_net).cpu().double()
    op = torch.optim.Adam([p for p in pol.parameters() if p.requires_grad], lr=5e-5)
    ov = torch.optim.Adam(val.parameters(), lr=3e-4)
    x, a = batch.states[:n].detach().cpu().double(), batch.actions[:n].detach().cpu().double()
    ret, adv = torch.randn(n, 1, dtype=torch.float64), torch.randn(n, 1, dtype=torch.float64)
    with torch.no_grad():
        flp = pol.get_log_prob(x, a)

    def epoch():
        vl = (val(x) - ret).pow(2).mean()
        ov.zero_grad(); vl.backward(); ov.step()
        ratio = torch.exp(pol.get_log_prob(x, a) - flp)
        pl = -torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).mean()
        op.zero_grad(); pl.backward(); op.step()

    epoch()  # warm-up (thread pool, allocator)
    ts = []
    t_all = time.perf_counter()
    while len(ts) < 3 or (time.perf_counter() - t_all < 8.0 and len(ts) < 10):
        t0 = time.perf_counter()
        epoch()
        ts.append(time.perf_counter() - t0)
    el = float(np.median(ts))
    return {"ppo_value": n / (10 * el), "ppo_unit": "samples/s (10-epoch full-batch update)",
            "ppo_sample": f"median of {len(ts)} full epochs over {n} samples, torch CPU float64, {torch.get_num_threads()} threads (physical cores)"}


GENERATED_CLASS = {"mesh": True, "model": "smpl"}  # a reference config's robot block: Robot() then emits body-body collisions, the excludes and rel_joint_lm ranges


def build_agent(args, rank, local, dtype, shapes=0, robot_cfg="default", cfg_over=None, objects=0, envs=None):
    """AgentCopycat on synthetic clips for the copycat rollout: `shapes` body shapes (configs[3]); `robot_cfg` = the config's robot block
    ("default": the model class the reference generates -- body-body collisions on, rel_joint_lm ranges --, or with --floor-only the shipped
    static asset as config/uhc_amd/copycat_mi355x.yml has it; None: that yml; a dict: a reference config's own keys); `objects` = K free
    5 kg boxes of 0.3 m behind every humanoid with synthetic obj_pose clips (configs[4])."""
    import tempfile
    from uhc_amd import sim as S
    from uhc_amd.agents.agent_copycat import AgentCopycat
    from uhc_amd.data_loaders.dataset_amass_single import DatasetAMASSSingle
    from uhc_amd.data_loaders.synthetic import make_synthetic_amass
    from uhc_amd.utils.config_utils.copycat_config import Config

    cfg = Config(cfg_id="copycat_mi355x", base_dir=tempfile.mkdtemp(prefix="uhc_bench_"))
    cfg.n_env = envs or args.envs  # (envs: a probe at another batch size, e.g. configs[2]'s 4096 per GPU)
    if isinstance(robot_cfg, str):
        robot_cfg = None if getattr(args, "floor_only", False) else GENERATED_CLASS
    if robot_cfg is not None:
        cfg.robot_cfg = dict(robot_cfg)
    for k, v in (cfg_over or {}).items():  # config keys of another reference config (e.g. copycat_ball_1.yml's controller block)
        setattr(cfg, k, v)
        cfg.cfg_dict[k] = v
    if args.pgs_iterations:
        cfg.pgs_iterations = args.pgs_iterations
    if args.solver is not None:
        cfg.contact_solver = args.solver
    cfg.no_log = True
    specs = dict(cfg.data_specs)
    specs["file_path"] = "synthetic"
    torch.manual_seed(cfg.seed)
    np.random.seed(cfg.seed + rank)
    n_clips = max(args.clips, shapes + 1) if shapes else args.clips
    dl = DatasetAMASSSingle(specs, "train", pickle_data=make_synthetic_amass(n_clips, seed=1 + rank, objects=objects))
    shape_models = clip_model = None
    if shapes:  # per-body length scale s_b ~ U(0.85, 1.15) (mass ~ s^3, inertia ~ s^5), default_rng(7); one shape per clip
        from uhc_amd.model.mjcf import kinematics_np, quat_to_mat, scale_model_per_body
        base = S.load_asset_model()
        srng = np.random.default_rng(7)
        stand = np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz"))["qpos"]

        def lowest(m):  # lowest hull vertex of the standing pose: the clip of a longer-legged body carries its root higher
            xp, xq, _, _ = kinematics_np(m, stand)
            return min((m.mesh_vert[m.geom_vertadr[g]:m.geom_vertadr[g] + m.geom_vertnum[g]] @ quat_to_mat(xq[m.geom_bodyid[g]]).T + xp[m.geom_bodyid[g]])[:, 2].min()
                       for g in range(m.ngeom) if m.geom_type[g] == 7)

        z0 = lowest(base)
        shape_models, lift = [], [0.0]
        for _ in range(shapes):
            sm = scale_model_per_body(base, np.r_[1.0, srng.uniform(0.85, 1.15, size=base.nbody - 1)])
            shape_models.append(sm)
            lift.append(z0 - lowest(sm))
        keys = list(dl.data_keys)
        clip_model = {k: (i % (shapes + 1)) for i, k in enumerate(keys)}
        for k, mi in clip_model.items():
            dl.data["trans"][k] = dl.data["trans"][k] + np.array([0.0, 0.0, lift[mi]])
    obj = None
    if objects:  # K free boxes (0.3 m, 5 kg, contype 1) dropped around each humanoid
        from uhc_amd.model.shapes import box_triangles
        obj = dict(hulls=[box_triangles(0.15, 0.15, 0.15)] * objects, density=5.0 / 0.027, friction=1.0, condim=1)
    agent = AgentCopycat(cfg, dtype, torch.device("cuda", local), data_loader=dl, shape_models=shape_models, clip_model=clip_model, objects=obj)
    agent.logger.handlers = [h for h in agent.logger.handlers if not isinstance(h, __import__("logging").StreamHandler) or hasattr(h, "baseFilename")]
    return agent


def step_roofline(name, n_env, env_steps_per_s, kernel_ms, nq, nv, nbody, action_dim):
    """Roofline block of a probe.  ALU: the workload's f64 flop per env-step (all tiers' step kernels, committed counter pass) x the
    measured env-steps/s against the FP64 vector peak.  HBM: algorithmic state bytes per env-step x env-steps/s against the HBM peak."""
    algo = 8 * (2 * nq + 3 * nv + action_dim + 69 + 10 * nbody)  # read qpos, qvel, warm start, action, target; write qpos, qvel, qacc, xpos, xquat, xipos
    alu = alu_per_env_step(name)
    out = {"bound": "fp64_valu", "unit": "TFLOP/s", "peak": 78.6, "first_tier_kernel_ms": kernel_ms,
           "hbm": {"algorithmic_bytes_per_env_step": algo, "achieved_GBs": algo * env_steps_per_s / 1e9, "peak_GBs": HBM_PEAK_GBS, "frac": algo * env_steps_per_s / 1e9 / HBM_PEAK_GBS}}
    if alu:
        out.update({"achieved": alu["flop_per_env_step"] * env_steps_per_s / 1e12, "frac": alu["flop_per_env_step"] * env_steps_per_s / 78.6e12,
                    "flop_per_env_step": alu["flop_per_env_step"], "f64_share_of_valu_instructions": alu.get("f64_share_of_valu"), "source": alu["source"]})
        if alu.get("hbm_bytes_per_env_step") is not None:
            out["hbm"]["traffic_bytes_per_env_step"] = alu["hbm_bytes_per_env_step"]
            out["hbm"]["traffic_over_algorithmic"] = alu["hbm_bytes_per_env_step"] / algo
    else:
        out.update({"achieved": None, "frac": None, "source": f"no committed counter pass for this workload (profiles/*_alu_{name}.json)"})
    return out


def rollout_probe(args, local, dtype, name, warmup=None, steps=None, reps=None, key=None, **kw):
    """A rollout of a VARIANT of the metric's workload after the timed region: same step, same envs per GPU, own agent.  `reps`
    repetitions of `steps` timed steps after `warmup` untimed ones (the restart of all envs at the head of a pass is a transient of a
    few episode lengths); the sub-line is the MEDIAN repetition, every repetition is listed.  Constraint rows dropped beyond the last
    tier's capacity are counted per step on the device (UHC_F_REDO bit 7 through uhc_rollout_record), not read off sticky flags."""
    from uhc_amd import sim as S
    warmup = args.probe_warmup if warmup is None else warmup
    steps = args.probe_steps if steps is None else steps
    reps = args.probe_reps if reps is None else reps
    agent = build_agent(args, 0, local, dtype, **kw)
    agent.per_epoch_update(0)
    env = agent.env
    n_env = env.n_env
    agent.rollout_begin(warmup + reps * steps)
    for _ in range(warmup):
        agent.rollout_step()
    torch.cuda.synchronize()
    runs, hist = [], []
    nefc_max = ncon_max = 0
    for _ in range(reps):
        env.sim.kernel_time()
        env.sim.set_timing(True)
        redo0 = agent._ro.redo_counts.clone()
        t0 = time.perf_counter()
        for _ in range(steps):
            agent.rollout_step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        redo_d = (agent._ro.redo_counts - redo0).cpu().tolist()
        ms, k = env.sim.kernel_time()
        env.sim.set_timing(False)
        nefc, ncon = env.sim.field(S.F_NEFC).cpu().numpy(), env.sim.field(S.F_NCON).cpu().numpy()
        hist.append(nefc)
        nefc_max, ncon_max = max(nefc_max, int(nefc.max())), max(ncon_max, int(ncon.max()))
        runs.append({"env_steps_per_s": n_env * steps / el, "ms_per_step": 1e3 * el / steps, "first_tier_kernel_ms": ms / max(k, 1),
                     "general_or_large_tier_share_of_env_steps": redo_d[0] / (n_env * steps), "large_tier_share_of_env_steps": redo_d[3] / (n_env * steps),
                     "sweeps_fallback_share_of_env_steps": redo_d[1] / (n_env * steps), "windowed_exact_solve_share_of_env_steps": redo_d[4] / (n_env * steps),
                     "tier4_primal_newton_share_of_env_steps": redo_d[5] / (n_env * steps), "tier4_newton_hit_its_cap_env_steps": int(redo_d[6]),
                     "efc_overflow_env_steps": int(redo_d[2])})
    _, logger = agent.rollout_end()
    med = sorted(runs, key=lambda r: r["env_steps_per_s"])[len(runs) // 2]
    nefc = np.concatenate(hist)
    m = env.model
    out = dict(med)
    out.update({"steps": steps, "warmup": warmup, "reps": reps, "env_steps_per_s_each_rep": [r["env_steps_per_s"] for r in runs],
                "efc_overflow_env_steps_all_reps": int(sum(r["efc_overflow_env_steps"] for r in runs)),
                "nefc_mean": float(nefc.mean()), "nefc_max_at_rep_ends": nefc_max, "ncon_max_at_rep_ends": ncon_max,
                "nefc_hist_edges": [0, 1, 17, 33, 49, 65, 97, 129, 193, 257, 513, 1025], "nefc_hist": np.histogram(nefc, bins=[0, 1, 17, 33, 49, 65, 97, 129, 193, 257, 513, 1025])[0].tolist(),
                "failed_envs_now": int(env.sim.field(S.F_FAIL).sum().item()),
                "avg_episode_len": logger.avg_episode_len, "avg_reward": logger.avg_c_reward,
                "model": {"nq": int(m.nq), "nv": int(m.nv), "nbody": int(m.nbody), "objects": int(getattr(env, "num_obj", 0))},
                "roofline": step_roofline(key or name, n_env, med["env_steps_per_s"], med["first_tier_kernel_ms"], int(m.nq), int(m.nv), int(m.nbody), env.action_dim),
                "workload": name})
    env.close()
    del agent
    torch.cuda.empty_cache()
    return out


PROBES = {
    "configs2_per_gpu": dict(name="configs[2]'s share of one GPU: the headline's rollout step at 4096 envs per GPU (the config shards 8 x 4096 over 8 GPUs with no "
                                  "data-path collective: SURVEY 8e); 3 x 20 steps", envs=4096, steps=20),
    "floor_only": dict(name="configs[1] on the shipped static asset as it is: floor contacts only, +-180 degree joint ranges (the headline of rounds 1-3; "
                            "no reference config runs it -- their env model comes out of Robot(cfg.robot_cfg))", robot_cfg=None),
    "shapes": dict(name="configs[3] smpl_shape: 64 body shapes (per-body length scales ~ U(0.85, 1.15), default_rng(7)), one model blob per clip, on the generated model "
                        "class (`--shapes 1023` runs 1024 of them)", shapes=63),
    "ball_rollout": dict(name="configs[4]'s env without objects: config/copycat_ball/copycat_ball_1.yml's humanoid and controller -- ball joints (nq 99), body-body collisions "
                              "on, action_type torque (tq_mul 4), no residual force, reward world_rfc_implicit_quat, observation get_full_obs_v2_quat (534) -- as a rollout "
                              "through env + policy; same clips", robot_cfg={"mesh": True, "model": "smpl", "ball": True},
                         cfg_over=dict(action_type="torque", residual_force=False, meta_pd=False, meta_pd_joint=False, reward_id="world_rfc_implicit_quat", obs_v=2, tq_mul=4,
                                       env_init_noise=0.0)),
    "configs4": dict(name="configs[4]: copycat_ball config WITH object contacts as ONE rollout through env + policy: the ball-joint humanoid above + 4 free 5 kg boxes of 0.3 m "
                          "per env (SURVEY 8d-5 stand-in for the licensed GRAB objects), dropped around the humanoid at every reset from the clip's obj_pose "
                          "(uhc/envs/humanoid_im.py:1284-1287); observation / reward / termination read the humanoid (qpos[:qpos_lim])",
                     robot_cfg={"mesh": True, "model": "smpl", "ball": True}, objects=4,
                     cfg_over=dict(action_type="torque", residual_force=False, meta_pd=False, meta_pd_joint=False, reward_id="world_rfc_implicit_quat", obs_v=2, tq_mul=4,
                                   env_init_noise=0.0)),
}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the env, RCCL rendezvous on 127.0.0.1) and relay rank 0's JSON line.  Under
    `python -m torch.distributed.run` WORLD_SIZE is already set and this is skipped."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write("".join(l + "\n" for l in out.splitlines() if l.startswith("{")))  # the JSON line only (gloo chats on stdout)
    sys.stdout.flush()
    sys.exit(rc)


def bench_ball_objects(args):
    """BASELINE configs[4] stand-in (the real config needs licensed SMPL / GRAB files and the reference's ball-joint env
    path does not run as shipped): ball-joint humanoid (one ball joint + three gear-vector motors per bone, nq 99, action_type torque,
    tq_mul 4: config/copycat_ball/copycat_ball_1.yml), body-body collisions on, K free boxes (0.3 m, 5 kg, contype 1) dropped around every
    humanoid, default_rng(11); random torque actions, every env re-posed every 30 control steps.  Physics only (uhc_batch_simulate):
    the quaternion observation / reward of that config are not built.  One JSON line, value = env-steps/s of this GPU."""
    import dataclasses
    from uhc_amd.model.shapes import box_triangles
    from uhc_amd import sim as S
    from uhc_amd.model.mjcf import add_free_bodies, ball_variant, hinge_to_ball_qpos, self_collision_variant
    torch.cuda.set_device(0)
    base = S.load_asset_model()
    rng = np.random.default_rng(11)
    K = args.objects
    ang = rng.uniform(0, 2 * np.pi, size=K)
    poses = np.stack([np.r_[-0.15 + 0.75 * np.cos(a), -0.05 + 0.75 * np.sin(a), 0.3 + 0.45 * k, 1, 0, 0, 0] for k, a in enumerate(ang)]) if K else np.zeros((0, 7))
    hb = ball_variant(base, damping=5.0)  # joint damping inside copycat_ball_1.yml's [0, 10] range: mj_Euler's implicit-damping branch
    m = self_collision_variant(hb)
    if K:
        m = add_free_bodies(m, [box_triangles(0.15, 0.15, 0.15)] * K, poses, density=5.0 / 0.027)
    m = dataclasses.replace(m, solver=1 if args.solver is None else args.solver, iterations=args.pgs_iterations or 300)
    ctrl = S.make_ctrl(base, action_type="torque", residual_force=False, meta_pd=False, tq_mul=4)
    n_env = args.envs
    stand = np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz"))["qpos"]
    q0 = np.tile(m.qpos0, (n_env, 1))
    for e in range(n_env):
        qh = stand.copy()
        qh[7:] += rng.normal(scale=0.1, size=69)
        q0[e, :99] = hinge_to_ball_qpos(base, hb, qh)
    v0 = np.zeros((n_env, m.nv))
    v0[:, :75] = rng.normal(scale=0.2, size=(n_env, 75))
    sim = S.SimBatch(m, ctrl, n_env)
    sim.set_kernel_path(1 if args.general_only else (0 if args.fixed_path else 2))
    q0d, v0d = torch.from_numpy(q0).cuda(), torch.from_numpy(v0).cuda()
    tb = torch.zeros(n_env, 69, dtype=torch.float64, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(11)
    acts = 0.1 * torch.randn(16, n_env, ctrl.action_dim, dtype=torch.float64, device="cuda", generator=gen)  # init-policy noise (log_std -2.3): torques ~ N(0, 10 N m)
    hist_nefc, hist_ncon, redo_tot, steps_done = [], [], 0, 0
    hooks = getattr(args, "hooks", None)  # (tools/tier_trace.py: called around chosen control steps)

    def run(k, timed):
        nonlocal redo_tot, sweep_tot, big_tot, over_tot, steps_done
        for i in range(k):
            if steps_done % 30 == 0:
                sim.set_state(q0d, v0d)
            if hooks and timed:
                hooks[0](sim, i)
            sim.simulate(acts[steps_done % 16], tb)
            if hooks and timed:
                hooks[1](sim, i)
            steps_done += 1
            if timed:
                redo_tot += ((sim.field(S.F_REDO) & 1) != 0).int()  # device-side accumulation, no sync
                over_tot += ((sim.field(S.F_REDO) & 0x80) != 0).int()  # rows dropped beyond the last tier's capacity in THIS step (the sticky flag is cleared by the next set_state)
                big_tot += ((sim.field(S.F_REDO) & 0x40) != 0).int()  # computed by the large tier (> 128 rows / 64 contacts / 20 body-body rows)
                sweep_tot += ((sim.field(S.F_REDO) & 2) != 0).int()
                r = sim.field(S.F_REDO)
                for j in range(4):  # bits 2-5: why the working sets gave up (friction rows / > 64 candidates in one island / no convergence / unsolved set)
                    why_tot[j] += ((r >> (2 + j)) & 1).sum()
                sub_tot[0] += sum(((r >> (8 + j)) & 1).sum() for j in range(15))  # bits 8+: the substeps solved by sweeps
                if i % 5 == 4:
                    hist_nefc.append(sim.field(S.F_NEFC).clone()); hist_ncon.append(sim.field(S.F_NCON).clone())

    redo_tot = torch.zeros(n_env, dtype=torch.int32, device="cuda")
    sweep_tot = torch.zeros(n_env, dtype=torch.int32, device="cuda")
    big_tot = torch.zeros(n_env, dtype=torch.int32, device="cuda")
    over_tot = torch.zeros(n_env, dtype=torch.int32, device="cuda")
    why_tot = torch.zeros(4, dtype=torch.int64, device="cuda")
    sub_tot = torch.zeros(1, dtype=torch.int64, device="cuda")
    run(args.warmup, False)
    torch.cuda.synchronize()
    sim.set_timing(True)
    t0 = time.perf_counter()
    run(args.steps, True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms, k = sim.kernel_time()
    nefc = torch.cat(hist_nefc).cpu().numpy() if hist_nefc else np.zeros(1)
    redo_share = float(redo_tot.double().sum().item()) / (n_env * args.steps)
    ncon = torch.cat(hist_ncon).cpu().numpy() if hist_ncon else np.zeros(1)
    out = {"metric": "env-steps/sec (ball-joint SMPL humanoid + free objects, 15 substeps/step, physics only)", "value": n_env * args.steps / el, "unit": "env-steps/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"configs[4] stand-in: ball-joint humanoid (nq {m.nq}, nv {m.nv}), self-collision on, {K} free 5 kg boxes, torque actions, {n_env} envs, "
                                  "re-posed every 30 control steps; physics only", "envs_per_gpu": n_env, "objects": K, "kernel_path": "general only" if args.general_only else ("fast, then general on the envs beyond its capacity" if args.fixed_path else "adaptive (uhc_batch_set_kernel_path 2)"),
                      "contact_solver": "exact: active set in the fast kernel, working sets in the general kernel" if int(m.solver) == 1 else "pgs sweeps", "pgs_sweep_cap": int(m.iterations)},
           "roofline": {"bound": "hbm", "kernel": "uhc_step_kernel<0, 2, true> (general tier)" if (args.general_only or redo_share > 0.6) else "uhc_step_kernel<0, 1, true> (fast tier)", "kernel_ms": ms / max(k, 1), "launches": k,
                        "achieved": 8 * (2 * m.nq + 3 * m.nv + ctrl.action_dim + 7 * m.nbody) * n_env / (ms / max(k, 1) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "note": "HIP events around the first tier launched (the fast kernel, or the general tier when the adaptive path skips the fast one); envs beyond a "
                                "tier's capacity (fast: 64 rows / 16 contacts / 12 body-body rows; general: 128 / 64 / 20) are redone by the next tier in a further launch, "
                                "which ms_per_step includes"},
           "workload_stats": {"nefc_mean": float(nefc.mean()), "nefc_max": int(nefc.max()), "ncon_mean": float(ncon.mean()), "ncon_max": int(ncon.max()),
                              "nefc_hist_edges": [0, 1, 17, 33, 49, 65, 97, 129, 193, 257, 513, 1025], "nefc_hist": np.histogram(nefc, bins=[0, 1, 17, 33, 49, 65, 97, 129, 193, 257, 513, 1025])[0].tolist(),
                              "general_kernel_share_of_env_steps": float(redo_tot.double().sum().item()) / (n_env * args.steps),
                              "sweeps_fallback_share_of_env_steps": float(sweep_tot.double().sum().item()) / (n_env * args.steps),
                              "sweeps_fallback_share_of_substeps": float(sub_tot.item()) / (n_env * args.steps * 15),
                              "sweeps_fallback_reasons_env_steps": dict(zip(["friction_rows", "island_needs_over_64_rows", "no_convergence", "unsolved_working_set"], why_tot.cpu().tolist())),
                              "efc_overflow_envs": int(sim.field(S.F_EFC_OVERFLOW).sum().item()),
                              # (the flag above is sticky only until an env's next set_state -- every 30 steps here; this one counts every timed step)
                              "efc_overflow_env_steps": int(over_tot.sum().item()),
                              "failed_envs": int(sim.field(S.F_FAIL).sum().item())}}
    out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS
    out["workload_stats"]["large_tier_share_of_env_steps"] = float(big_tot.double().sum().item()) / (n_env * args.steps)
    sim.close()
    return out


def main():
    args = parse()
    if args.workload == "ball_objects":
        print(json.dumps(bench_ball_objects(args)))
        return
    if args.only_probe:  # one probe alone (profiling runs: rocprofv3 around this process sees that workload's kernels only)
        torch.cuda.set_device(0)
        torch.set_default_dtype(torch.float64)
        pr = dict(PROBES[args.only_probe])
        print(json.dumps(rollout_probe(args, 0, torch.float64, pr.pop("name"), key=args.only_probe, **pr)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    torch.cuda.set_device(local)
    if dist_on:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            td.init_process_group(args.backend, rank=rank, world_size=world)
    dtype = torch.float64 if args.ppo_dtype == "float64" else torch.float32
    torch.set_default_dtype(dtype)
    from uhc_amd import sim as S
    agent = build_agent(args, rank, local, dtype, shapes=args.shapes)
    agent.per_epoch_update(0)
    env = agent.env
    n_env = env.n_env
    T = args.preroll + args.warmup + args.steps

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            td.barrier()
            torch.cuda.synchronize()

    agent.rollout_begin(T)
    for _ in range(args.preroll + args.warmup):  # (pre-roll: past the restart transient; then the W warm-up steps the contract asks for)
        agent.rollout_step()
    fence()
    env.sim.set_timing(True)
    redo0 = agent._ro.redo_counts.clone()  # env-steps the general kernel had to redo: counted on the device by the step's own bookkeeping launch
    t0 = time.perf_counter()
    t_half = t0
    for k in range(args.steps):
        if k == args.steps // 2:
            t_half = time.perf_counter()  # (no synchronisation of its own: the host waits for every step's snapshot, so it is never more than one step ahead)
        agent.rollout_step()
    fence()
    elapsed = time.perf_counter() - t0
    half_ms = (1e3 * (t_half - t0) / max(1, args.steps // 2), 1e3 * (t0 + elapsed - t_half) / max(1, args.steps - args.steps // 2))
    redo_d = (agent._ro.redo_counts - redo0).cpu().tolist()
    kern_total_ms, kern_n = env.sim.kernel_time()
    env.sim.set_timing(False)
    nefc = env.sim.field(S.F_NEFC).cpu().numpy()
    ncon = env.sim.field(S.F_NCON).cpu().numpy()
    iters = env.sim.field(S.F_SOLVER_ITER).cpu().numpy()
    overflow = int(env.sim.field(S.F_EFC_OVERFLOW).sum().item())
    batch, logger = agent.rollout_end()
    per_rank = [n_env * args.steps / elapsed]
    if dist_on:
        tt = torch.tensor([elapsed], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        allt = [torch.empty_like(tt) for _ in range(world)]
        td.all_gather(allt, tt)
        per_rank = [n_env * args.steps / float(x.item()) for x in allt]
        elapsed = max(float(x.item()) for x in allt)  # the job is as fast as its slowest rank
    # ---- one PPO update over the collected samples (outside the timed region of `value`)
    ppo = None
    if not args.no_ppo:
        agent.update_params(batch)  # untimed: the first update of a process pays rocBLAS kernel loading and allocator growth (2x)
        agent.comm_summary()
        agent.time_comm = dist_on
        fence()
        t1 = time.perf_counter()
        agent.update_params(batch)
        fence()
        t_up = time.perf_counter() - t1
        ncalls, comm_ms, comm_bytes = agent.comm_summary()
        if dist_on:
            tt = torch.tensor([t_up], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
            td.all_reduce(tt, op=td.ReduceOp.MAX)
            t_up = float(tt.item())
        n_samples = batch.states.shape[0] * world
        # update_params alone: value forward (7.93 M) + fixed-log-prob forward (8.04 M) + 10 epochs x 3 x (8.04 + 7.93 M) = 495 MFLOP per sample
        # (SURVEY 8d's 503 M per iteration includes the rollout's policy forward, which the timed update does not perform: VERDICT r4 weak 11)
        flops = n_samples * (7.934976e6 + 8.041472e6 + 10 * 3 * (8.041472e6 + 7.934976e6))
        ppo = {"samples": n_samples, "update_s": t_up, "samples_per_s": n_samples / t_up, "dtype": args.ppo_dtype, "epochs": agent.cfg.num_optim_epoch,
               "gemm_tflops": flops / t_up / 1e12, "mfma_util": flops / t_up / 1e12 / (78.6 if args.ppo_dtype == "float64" else 157.3),
               "mfma_peak_tflops": 78.6 if args.ppo_dtype == "float64" else 157.3, "rollout_plus_update_samples_per_s": n_samples / (t_up + elapsed * T / args.steps)}
        if ncalls:  # rank 0's view of the gradient exchange: one flat all-reduce per network per optimisation step
            algbw = comm_bytes * ncalls / (comm_ms * 1e-3) / 1e9
            ovl = bool(getattr(agent, "overlap_grad_exchange", True))
            ppo["allreduce"] = {"calls": ncalls, "bytes_per_call": comm_bytes, "total_ms": comm_ms, "algbw_GBs": algbw,
                                "busbw_GBs": algbw * 2 * (world - 1) / world, "share_of_update": comm_ms * 1e-3 / t_up,
                                # value half first, the surrogate's backward pass enqueued behind it: an interval (start of an exchange -> its
                                # wait over) then contains that backward pass, the bandwidths above are lower bounds, and what the update pays is
                                "overlapped_with_policy_backward": ovl, "exposed_ms": agent.comm_exposed_ms,
                                "exposed_share_of_update": agent.comm_exposed_ms * 1e-3 / t_up,
                                # 1 - exposed / total: the share of the exchanges' wall time (start of the collective -> end of the wait) during which the compute stream was NOT
                                # standing still -- over RCCL the answer to "does the all-reduce overlap the surrogate's backward pass" (over gloo there is no side stream: ~0)
                                "hidden_share_of_exchange_time": (1.0 - agent.comm_exposed_ms / comm_ms) if comm_ms > 0 else None}
    # ---- the same rollout with MuJoCo-style PGS sweeps (solver 0), kernel time only: what north_star's "PGS contact solve" costs
    pgs = None
    if world == 1 and not args.no_pgs_probe:
        pgs = {}
        orig = (int(env.model.solver), int(env.model.iterations))
        for cap in (100, 300):
            env.sim.set_solver(0, cap)
            agent.rollout_begin(12)
            for _ in range(4):
                agent.rollout_step()
            torch.cuda.synchronize()
            env.sim.kernel_time()
            env.sim.set_timing(True)
            for _ in range(8):
                agent.rollout_step()
            torch.cuda.synchronize()
            ms, k = env.sim.kernel_time()
            env.sim.set_timing(False)
            sweeps = float(env.sim.field(S.F_SOLVER_ITER).double().mean().item())
            agent.rollout_end()
            pgs[f"sweep_cap_{cap}"] = {"kernel_ms": ms / max(k, 1), "kernel_only_env_steps_per_s": n_env / (ms / max(k, 1) * 1e-3), "mean_sweeps_last_substep": sweeps, "launches": k}
        env.sim.set_solver(*orig)
    # ---- the other configs of BASELINE.json as probes of the same rollout step (own agent each; median of `--probe-reps` repetitions)
    probes = {}
    if world == 1 and not args.no_probes and not args.shapes:
        for key, pr in PROBES.items():
            if key == "floor_only" and args.floor_only:
                continue  # (it is the headline of this run)
            pr = dict(pr)
            probes[key] = rollout_probe(args, local, dtype, pr.pop("name"), key=key, **pr)
        if args.floor_only:  # the reference's model class as a sub-line when the headline is the static asset
            probes["self_collision"] = rollout_probe(args, local, dtype, "configs[1] on the model class the reference generates (body-body collisions on, rel_joint_lm ranges)",
                                                     key="self_collision", robot_cfg=GENERATED_CLASS)
    if rank == 0:
        kern_ms = max(kern_total_ms / max(kern_n, 1), 1e-9)
        dense = int(env.model.geom_contype[1:].sum()) > 0  # body-body contacts compiled in: the <0, 1, true> instantiation
        kname = FAST_KERNEL[dense]
        tag = "" if not dense else "_selfcol"  # the committed counter passes of the self-colliding workload carry this tag
        hbm_achieved = ALGO_BYTES_PER_ENV_STEP * n_env / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(kname, tag) if n_env == 1024 else (None, None)
        # the bound that matters: float64 vector issue / latency.  Flops from the PMC instruction counters of the same workload when a
        # committed pass exists (profiles/*_pmc_VALU_F64<tag>.txt, 1024 envs), else the SURVEY 8d estimate of ~20 MFLOP per env-step.
        # Under sticky tiers the fast tier's launch does not compute every env of the step (a few per cent run in the general / large tier
        # beside it), and the counters are per launch of THIS kernel: flops and time belong to the same launches.
        vc = valu_f64_counters(kname, tag) if n_env == 1024 else None
        flop_launch = vc["flop_per_launch"] if vc else 20e6 * n_env
        tflops = flop_launch / (kern_ms * 1e-3) / 1e12
        # SURVEY 8d's stand-alone contact-solve accounting: A (nefc^2) + b, R, f in + f out = 8 (nefc^2 + 4 nefc) bytes per solve, 15 solves per
        # launch; the time is the kernel's contact-solve share (stage profile).  The solve keeps A in registers: it is nowhere near HBM-bound.
        share, share_src = contact_solve_share(tag)
        cs_bytes = float((8.0 * (nefc.astype(np.float64) ** 2 + 4.0 * nefc)).sum()) * 15
        csolve = {"algorithmic_bytes_per_launch": cs_bytes, "share_of_kernel": share, "share_source": share_src,
                  "achieved_GBs": (cs_bytes / (kern_ms * 1e-3 * share) / 1e9) if share else None,
                  "frac_of_hbm_peak": (cs_bytes / (kern_ms * 1e-3 * share) / 1e9 / HBM_PEAK_GBS) if share else None,
                  "formula": "sum over envs of 8 (nefc^2 + 4 nefc) x 15 substeps, nefc of the last substep"}
        whole = alu_per_env_step("floor_only" if not dense else "headline")
        model_class = ("the model class the reference's env runs: what Robot(cfg.robot_cfg) generates -- body-body collisions on, Chest / shoulder excludes, rel_joint_lm "
                       "joint ranges" if dense else "the shipped static asset as it is (floor contacts only; --floor-only)")
        out = {
            "metric": "env-steps/sec (69-DoF SMPL humanoid, 15 substeps/step)", "value": n_env * args.steps * world / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "per_rank_env_steps_per_s": per_rank, "steps": args.steps, "warmup": args.warmup, "preroll": args.preroll, "ms_per_step": 1e3 * elapsed / max(1, args.steps),
            # steady state: the two halves of the timed region, on rank 0's clock -- a timed region still inside the restart transient shows as a drift
            "steady_state": {"first_half_ms_per_step": half_ms[0], "second_half_ms_per_step": half_ms[1], "drift": half_ms[1] / half_ms[0] - 1.0 if half_ms[0] > 0 else None},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: copycat rollout step (obs filter, policy MLP 657-2048-1024-512-105 sampling, PD target, fused "
                                   f"physics, termination, reward, obs v2, resets), {n_env} batched envs/GPU, {args.clips} synthetic clips/rank, "
                                   f"random-init policy, on {model_class}", "envs_per_gpu": n_env, "substeps": 15, "body_body_collisions": bool(dense),
                       "contact_solver": ("exact optimum: active set on the dual QP in registers (fast tier), working sets of <= 64 rows (general / large tier), Newton on the primal problem for what those cannot hold or finish (tier 4: > 256 rows, islands with > 64 force-carrying rows)" if int(env.model.solver) == 1 else "pgs sweeps"), "pgs_sweep_cap": int(env.model.iterations), "body_shapes": 1 + args.shapes,
                       "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "fp64_valu", "kernel": "uhc_step_kernel<0, 2, true> (general tier)" if os.environ.get("UHC_FORCE_GENERAL") == "1" else kname[5:] + (" (fast tier, body-body contacts compiled in)" if dense else " (fast tier, floor-only model)"),
                         "achieved": tflops, "peak": 78.6, "unit": "TFLOP/s", "frac": tflops / 78.6,
                         "flop_per_launch": flop_launch, "flop_source": (vc["source"] + ": 64 x (ADD + MUL + TRANS + 2 FMA) f64 wave-instructions of this kernel, per launch") if vc else "estimate (SURVEY 8d: ~20 MFLOP per env-step)",
                         "counters_per_launch": vc["counters"] if vc else None,
                         # `frac` counts every f64 wave-instruction at 64 lanes (an upper bound); the same with the counters' average of lanes switched on per VALU cycle:
                         "active_lanes_per_valu_cycle": vc["active_lanes_per_valu_cycle"] if vc else None,
                         "flop_per_launch_active_lanes": vc["flop_per_launch_active_lanes"] if vc else None,
                         "frac_active_lanes": (vc["flop_per_launch_active_lanes"] / (kern_ms * 1e-3) / 1e12 / 78.6) if vc and vc["flop_per_launch_active_lanes"] else None,
                         "kernel_ms": kern_ms, "launches": kern_n, "kernel_only_env_steps_per_s": n_env / (kern_ms * 1e-3),
                         "rocprofv3_avg_ms": rocprof_avg_ms(kname, tag)[0] if n_env == 1024 else None, "rocprofv3_source": rocprof_avg_ms(kname, tag)[1] if n_env == 1024 else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "hbm": {"bound": "hbm", "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP, "traffic": traffic, "traffic_over_algorithmic": (traffic / (ALGO_BYTES_PER_ENV_STEP * n_env)) if traffic else None},
                         "whole_step": step_roofline("floor_only" if not dense else "headline", n_env, n_env * args.steps / elapsed, kern_ms, int(env.model.nq), int(env.model.nv), int(env.model.nbody), env.action_dim) if world == 1 else None,
                         "contact_solve": csolve,
                         "note": "fused f64 step, one env per wavefront: the state crosses HBM once per 15 substeps, so the kernel is bound by dependent f64 VALU / LDS / readlane "
                                 "latency with one wave per SIMD, not by HBM (DESIGN.md section 5): `frac` is the share of the 78.6 TFLOP/s FP64 vector peak, the HBM view is the "
                                 "`hbm` sub-block; traffic above the algorithmic bytes is L2 misses of the schedule tables / kernel code and register spills to scratch"},
            "workload_stats": {"nefc_mean": float(nefc.mean()), "nefc_max": int(nefc.max()), "solver_iters_mean": float(iters.mean()), "general_kernel_envs_last_step": int((env.sim.field(S.F_REDO) != 0).sum().item()),
                               "general_or_large_tier_env_steps_timed_region": int(redo_d[0]), "large_tier_env_steps_timed_region": int(redo_d[3]), "sweeps_fallback_env_steps_timed_region": int(redo_d[1]),
                               "efc_overflow_env_steps_timed_region": int(redo_d[2]), "windowed_exact_solve_env_steps_timed_region": int(redo_d[4]),
                               "tier4_primal_newton_env_steps_timed_region": int(redo_d[5]), "tier4_newton_hit_its_cap_env_steps_timed_region": int(redo_d[6]),
                               "nefc_hist_edges": [0, 1, 9, 17, 25, 33, 41, 49, 57, 65, 97, 129, 257, 1025],
                               "nefc_hist": np.histogram(nefc, bins=[0, 1, 9, 17, 25, 33, 41, 49, 57, 65, 97, 129, 257, 1025])[0].tolist(),
                               "ncon_hist_edges": [0, 1, 3, 5, 9, 13, 17, 33, 65], "ncon_hist": np.histogram(ncon, bins=[0, 1, 3, 5, 9, 13, 17, 33, 65])[0].tolist(),
                               "efc_overflow_envs_sticky_flags": overflow, "episodes": logger.num_episodes, "avg_episode_len": logger.avg_episode_len,
                               "avg_reward": logger.avg_c_reward},
        }
        if pgs:
            out["pgs"] = pgs
        out.update(probes)
        if ppo:
            out["ppo"] = ppo
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(agent, n_env)
            if ppo:
                out["cpu_baseline"].update(cpu_ppo_baseline(agent, batch))
        print(json.dumps(out))
    if dist_on:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
