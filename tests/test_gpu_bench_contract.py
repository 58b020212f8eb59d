"""GPU: bench.py prints ONE JSON line with the keys the driver and the judge read (task contract), on a tiny workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--preroll", "3", "--envs", "64", "--clips", "8"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "env-steps/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["value"] == pytest.approx(64 * 4 / (d["ms_per_step"] * 4e-3), rel=1e-6)
    assert "workload" in d["config"] and "model" not in d["config"]
    # the dominant kernel is FP64-vector bound: the top-level block is priced against the FP64 vector peak, the HBM
    # figures (algorithmic bytes per launch / the kernel's launch time, and the PMC traffic) sit in its "hbm" sub-block
    r = d["roofline"]
    assert r["bound"] == "fp64_valu" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6
    assert r["kernel_ms"] > 0 and r["launches"] == 4 and (r["frac"] is None or r["frac"] == pytest.approx(r["achieved"] / r["peak"]))
    h = r["hbm"]
    assert h["bound"] == "hbm" and h["unit"] == "GB/s" and h["peak"] == 8000.0 and h["frac"] == pytest.approx(h["achieved"] / h["peak"])
    assert h["achieved"] == pytest.approx(7008 * 64 / (r["kernel_ms"] * 1e-3) / 1e9, rel=1e-9)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "env-steps/s" and "sample" in c
    assert d["preroll"] == 3 and d["ppo"]["samples"] == 64 * 9 and d["ppo"]["samples_per_s"] > 0  # (pre-roll + warm-up + timed steps are one sampling pass)
    assert 0.0 < d["ppo"]["mfma_util"] == pytest.approx(d["ppo"]["gemm_tflops"] / 78.6)
    assert 1 <= c["cores"] <= len(os.sched_getaffinity(0)) == c["sched_affinity"] and c["scaling_efficiency"] == pytest.approx(c["value"] / (c["cores"] * c["value_1_thread"]))
    # the other BASELINE configs ride along as short probes of the same step: driver-visible lines, not builder-run extras
    for k in ("floor_only", "shapes", "ball_rollout", "configs4"):
        assert d[k]["env_steps_per_s"] > 0 and len(d[k]["env_steps_per_s_each_rep"]) == d[k]["reps"] and "workload" in d[k] and d[k]["roofline"]["bound"] == "fp64_valu", k
    assert d["floor_only"]["efc_overflow_env_steps_all_reps"] == 0 and d["shapes"]["efc_overflow_env_steps_all_reps"] == 0
    assert d["configs4"]["model"]["objects"] == 4 and d["configs4"]["model"]["nq"] == 99 + 28
    assert d["workload_stats"]["nefc_mean"] >= d["floor_only"]["nefc_mean"] - 8  # (the headline's body-body rows add to the floor rows)


def test_headline_is_measured_at_steady_state():
    """with the default pre-roll (40 untimed steps, two episode lengths of the random-init policy) the timed region is past the transient
    of the common restart: its two halves agree (15 %: 25 steps of a stochastic episode turnover each; inside the transient the first half is 20-40 % off),
    and the fast tier's HIP-event time per launch is within 35 % of the committed kernel trace (boxes of the pool differ by up to a quarter on the same binary)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "50", "--warmup", "10", "--no-probes", "--no-cpu-baseline", "--no-ppo", "--no-pgs-probe"],
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    ss = d["steady_state"]
    print(f"headline {d['value']:.0f} env-steps/s, {d['ms_per_step']:.2f} ms per step; halves {ss['first_half_ms_per_step']:.2f} / {ss['second_half_ms_per_step']:.2f} ms; "
          f"kernel {d['roofline']['kernel_ms']:.2f} ms, committed trace {d['roofline'].get('rocprofv3_avg_ms')}")
    assert d["preroll"] == 40 and abs(ss["drift"]) < 0.15, ss
    if d["roofline"].get("rocprofv3_avg_ms"):
        assert abs(d["roofline"]["kernel_ms"] / d["roofline"]["rocprofv3_avg_ms"] - 1.0) < 0.35, d["roofline"]



def test_bench_two_ranks_over_rccl():
    """Multi-GPU readiness: `bench.py --gpus 2` over RCCL (backend nccl), one rank per GPU -- runs wherever two devices are visible and
    skips on the one-GPU test box.  The gradient all-reduce is then a real xGMI exchange and its bus bandwidth is reported."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the round-end 8-GPU node); the one-GPU box runs the gloo variant below")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--preroll", "0", "--envs", "256", "--clips", "8", "--no-probes"],
                         cwd=ROOT, capture_output=True, text=True, timeout=1200,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    assert d["n_gpus"] == 2 and len(d["per_rank_env_steps_per_s"]) == 2
    ar = d["ppo"]["allreduce"]  # (two halves per epoch: the value gradient's travels during the surrogate's backward pass)
    assert ar["calls"] == 20 and ar["busbw_GBs"] > 0 and ar["overlapped_with_policy_backward"] and 0 <= ar["exposed_ms"] <= ar["total_ms"]
    # the first two-GPU box answers the open question (does RCCL's stream overlap the surrogate's backward pass?) in this one run: the record is kept
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_overlap.json"), "w") as f:
        json.dump({"allreduce": ar, "hidden_share_of_exchange_time": ar["hidden_share_of_exchange_time"]}, f)
    print("RCCL gradient exchange:", json.dumps(ar))


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself and reports the whole job.  On this one-GPU box both ranks
    share GPU 0, which RCCL refuses (duplicate device), so the plumbing check rides on gloo; on an N-GPU node the default backend is RCCL."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo", "--steps", "4", "--warmup", "2", "--preroll", "0", "--envs", "64",
                          "--clips", "8"], cwd=ROOT, capture_output=True, text=True, timeout=1200,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["per_rank_env_steps_per_s"]) == 2
    assert d["value"] == pytest.approx(2 * 64 * 4 / (d["ms_per_step"] * 4e-3), rel=1e-6)
    assert d["value"] <= sum(d["per_rank_env_steps_per_s"]) * (1 + 1e-9)
    ar = d["ppo"]["allreduce"]  # (per epoch: the value gradient's half, started before the surrogate's backward pass, and the policy's half)
    assert d["ppo"]["samples"] == 2 * 64 * 6 and ar["calls"] == 20 and ar["busbw_GBs"] > 0 and ar["overlapped_with_policy_backward"]
    assert 0 <= ar["exposed_ms"] <= ar["total_ms"] * 1.001, ar
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
