"""bench.py -- hot-path throughput on MI355X (contract: see the task statement / DESIGN.md section 6).

A "step" is one control step (15 physics substeps with stable-PD + residual force, then observation,
reward and termination when the env layer is present) of EVERY environment of the batch:
BASELINE.json configs[1] = 1024 batched envs per GPU, synthetic clips.  value = env-steps/s summed
over all ranks.  One process per GPU; envs shard across ranks with no data-path collective
(scaling = weak).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 8 * (76 + 75 + 75 + 105 + 69 + 76 + 75 + 75 + 75 + 100 + 75)
HBM_PEAK_GBS = 8000.0
EPISODE_LEN = 30  # synthetic episode length before an env is re-initialised (untrained policy falls in ~1 s)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=60)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--envs", type=int, default=1024, help="environments per GPU")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-envs", type=int, default=0, help="envs in the CPU baseline sample (0 = 4 x cores)")
    return p.parse_args()


def make_inputs(model, ctrl, n_env, seed):
    z = np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz"))
    rng = np.random.default_rng(seed)
    qpos = np.tile(z["qpos"], (n_env, 1))
    qpos[:, 7:] += rng.normal(scale=0.05, size=(n_env, model.nu))
    yaw = rng.uniform(-np.pi, np.pi, size=n_env)  # random heading about world z (left-multiplied)
    qz = np.stack([np.cos(yaw / 2), np.zeros(n_env), np.zeros(n_env), np.sin(yaw / 2)], axis=1)
    q = qpos[:, 3:7].copy()
    qpos[:, 3] = qz[:, 0] * q[:, 0] - qz[:, 3] * q[:, 3]
    qpos[:, 4] = qz[:, 0] * q[:, 1] - qz[:, 3] * q[:, 2]
    qpos[:, 5] = qz[:, 0] * q[:, 2] + qz[:, 3] * q[:, 1]
    qpos[:, 6] = qz[:, 0] * q[:, 3] + qz[:, 3] * q[:, 0]
    qvel = rng.normal(scale=0.1, size=(n_env, model.nv))
    # policy at initialisation: zero mean, std = exp(-2.3) (uhc_implicit_shape.yml:23)
    actions = rng.normal(scale=np.exp(-2.3), size=(8, n_env, ctrl.action_dim))
    return qpos, qvel, actions


def cpu_baseline(model, ctrl, qpos, qvel, actions, n_cpu_env):
    """The CPU oracle (own restatement of the MuJoCo step; 'port') on the host cores, bounded sample."""
    import ctypes as C
    from oracle.physics import OracleSim, lib
    cores = os.cpu_count() or 1
    n = n_cpu_env or min(len(qpos), 4 * cores)
    sims = [OracleSim(model, ctrl) for _ in range(n)]
    for e, s in enumerate(sims):
        s.set_state(qpos[e], qvel[e])
    L = lib()
    ptrs = (C.c_void_p * n)(*[s.d for s in sims])
    tb = np.ascontiguousarray(qpos[:n, 7:])
    steps = 0
    t0 = time.perf_counter()
    while True:
        a = np.ascontiguousarray(actions[steps % len(actions), :n])
        L.orc_batch_do_simulation(C.byref(sims[0].desc), C.byref(ctrl), ptrs, n, a.ctypes.data_as(C.POINTER(C.c_double)),
                                  tb.ctypes.data_as(C.POINTER(C.c_double)))
        steps += 1
        el = time.perf_counter() - t0
        if el > 12.0 or steps >= 20:
            break
    return {"value": n * steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} envs x {steps} control steps of the same workload (oracle/physics_oracle.c, OpenMP over envs)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1
    if dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    from uhc_amd import sim as S
    model = S.load_asset_model()
    ctrl = S.make_ctrl(model)
    n_env = args.envs
    qpos, qvel, actions = make_inputs(model, ctrl, n_env, seed=1 + rank)

    batch = S.SimBatch(model, ctrl, n_env, device=local)
    d_qpos, d_qvel = torch.from_numpy(qpos).cuda(), torch.from_numpy(qvel).cuda()
    d_act = torch.from_numpy(actions).cuda()
    d_tb = torch.from_numpy(np.ascontiguousarray(qpos[:, 7:])).cuda()
    batch.set_state(d_qpos, d_qvel)
    # staggered synthetic episodes: env e is re-initialised whenever (t + e) % EPISODE_LEN == 0
    phase = torch.arange(n_env, device="cuda") % EPISODE_LEN
    reset_ids = [torch.nonzero(phase == k).flatten().to(torch.int32) for k in range(EPISODE_LEN)]

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    nefc_hist = []

    def one_step(t, timed_idx=None):
        ids = reset_ids[t % EPISODE_LEN]
        if t > 0 and len(ids):
            batch.set_state(d_qpos[ids.long()], d_qvel[ids.long()], ids)
        a = d_act[t % d_act.shape[0]]
        if timed_idx is not None:
            ev[timed_idx][0].record()
        batch.simulate(a, d_tb)
        if timed_idx is not None:
            ev[timed_idx][1].record()

    def fence():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
            torch.cuda.synchronize()

    for t in range(args.warmup):
        one_step(t)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(args.warmup + k, k)
    fence()
    elapsed = time.perf_counter() - t0
    nefc_hist = batch.field(S.F_NEFC).cpu().numpy()
    iters = batch.field(S.F_SOLVER_ITER).cpu().numpy()
    fails = int(batch.field(S.F_FAIL).sum().item())
    overflow = int(batch.field(S.F_EFC_OVERFLOW).sum().item())
    if dist:
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else float("nan")
    total_env_steps = n_env * args.steps * world
    value = total_env_steps / elapsed
    if rank == 0:
        achieved = ALGO_BYTES_PER_ENV_STEP * n_env / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec (69-DoF SMPL humanoid, 15 substeps/step)", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: copycat (uhc_implicit_shape) control step, {n_env} batched envs/GPU, "
                                   "standing_neutral-derived synthetic clips, staggered 30-step episodes, init-policy action noise",
                       "envs_per_gpu": n_env, "substeps": 15, "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "kernel": "uhc_step_kernel<0>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kern_ms,
                         "note": "fused f64 step is latency/VALU bound, not HBM bound (DESIGN.md section 5)"},
            "workload_stats": {"nefc_mean": float(nefc_hist.mean()), "nefc_max": int(nefc_hist.max()),
                               "pgs_iters_mean": float(iters.mean()), "failed_envs": fails, "efc_overflow_envs": overflow},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, ctrl, qpos, qvel, actions, args.cpu_envs)
        print(json.dumps(out))
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
