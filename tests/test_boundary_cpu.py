"""CPU: the reference's import surface resolves (the `uhc` alias package), with the reference's registry keys and call signatures."""
import inspect


def test_reference_import_paths_resolve_to_this_build():
    import uhc_amd.agents
    import uhc_amd.agents.agent_copycat as real
    from uhc.agents import agent_dict                     # scripts/train_uhc.py:32
    from uhc.agents.agent_copycat import AgentCopycat     # scripts/train_uhc.py:90
    from uhc.data_loaders.dataset_amass_single import DatasetAMASSSingle
    from uhc.envs import env_dict
    from uhc.khrylib.rl.core import PolicyGaussian, Value, estimate_advantages  # noqa: F401
    from uhc.khrylib.utils.zfilter import ZFilter  # noqa: F401
    from uhc.smpllib.smpl_mujoco import SMPLConverter, smpl_to_qpose  # noqa: F401
    from uhc.utils.config_utils.copycat_config import Config  # scripts/train_uhc.py:31
    from uhc.utils.flags import flags  # noqa: F401
    assert agent_dict is uhc_amd.agents.agent_dict and AgentCopycat is real.AgentCopycat
    assert set(agent_dict) == {"agent_copycat"} and set(env_dict) == {"humanoid_im"}
    # constructor / method signatures the reference's callers rely on (agent_copycat.py:55, humanoid_im.py:49, dataset_amass_single.py)
    assert list(inspect.signature(AgentCopycat.__init__).parameters)[:6] == ["self", "cfg", "dtype", "device", "training", "checkpoint_epoch"]
    assert list(inspect.signature(env_dict["humanoid_im"].__init__).parameters)[:6] == ["self", "cfg", "init_expert", "data_specs", "mode", "no_root"]
    assert list(inspect.signature(DatasetAMASSSingle.sample_seq).parameters)[:6] == ["self", "full_sample", "freq_dict", "sampling_temp", "sampling_freq", "precision_mode"]
    for name in ("reset", "step", "load_expert", "set_mode", "seed", "get_expert_attr", "get_expert_index", "get_wbody_pos", "get_world_vf", "fail_safe",
                 "get_ee_pos", "get_body_quat", "get_com", "get_humanoid_qpos"):
        assert callable(getattr(env_dict["humanoid_im"], name)), name
    assert Config.__module__ == "uhc_amd.utils.config_utils.copycat_config"
    import pytest
    with pytest.raises(ModuleNotFoundError):
        import uhc.no_such_module  # noqa: F401


def test_bench_defaults_are_the_drivers_contract():
    """`python bench.py` with no flags: one GPU, a step count that finishes in minutes, the metric's configuration (1 024 envs, exact
    contact solve, float64 learner); the flags the driver passes exist."""
    import importlib.util
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_cli", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = argv
    assert a.gpus == 1 and a.steps == 50 and a.warmup == 10 and a.envs == 1024 and a.workload == "copycat" and a.ppo_dtype == "float64"
    assert a.solver is None and not a.general_only and not a.fixed_path and a.shapes == 0
    sys.argv = ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
    try:
        b = bench.parse()
    finally:
        sys.argv = argv
    assert (b.gpus, b.steps, b.warmup) == (8, 20, 5)


def test_bench_reads_every_committed_profile():
    """bench.py quotes two numbers from the committed profiles (contact-solve share of the stage profile, float64 VALU counters of the PMC
    pass); whatever a measurement pass leaves under profiles/ must parse -- a profile format change may not break the bench line."""
    import glob
    import importlib.util
    import os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_cli2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    share, src = bench.contact_solve_share()
    assert src and 0.02 < share < 0.6
    c = bench.valu_f64_counters(bench.FAST_KERNEL[False])
    assert c and c["flop_per_launch"] > 1e9
    t, tsrc = bench.pmc_traffic(bench.FAST_KERNEL[False])
    ms, msrc = bench.rocprof_avg_ms(bench.FAST_KERNEL[True], "_selfcol")  # the kernel trace of the headline workload, quoted beside the HIP-event time
    assert msrc and 1.0 < ms < 50.0 and bench.rocprof_avg_ms("no such kernel") == (None, None)
    assert tsrc and t > 1e6
    for name in ("headline", "floor_only", "shapes", "configs4", "ball_rollout"):  # per-workload counter passes (tools/pmc_alu.py), where committed
        a = bench.alu_per_env_step(name)
        assert a is None or (a["flop_per_env_step"] > 1e6 and a["source"].endswith(".json")), name
    # the probe at configs[2]'s batch size is the headline's workload: it is priced with the headline's flop per env-step
    assert (bench.alu_per_env_step("configs2_per_gpu") or {}).get("source") == (bench.alu_per_env_step("headline") or {}).get("source")
    # every BASELINE config that fits one GPU has a probe, and the probe of configs[2] runs at its 4096 envs per GPU
    assert set(bench.PROBES) == {"configs2_per_gpu", "floor_only", "shapes", "ball_rollout", "configs4"} and bench.PROBES["configs2_per_gpu"]["envs"] == 4096
    assert bench.PROBES["configs4"]["objects"] == 4 and bench.PROBES["configs4"]["robot_cfg"]["ball"] and bench.PROBES["ball_rollout"]["robot_cfg"]["ball"]
    r = bench.step_roofline("no_such_workload", 1024, 1e5, 3.5, 76, 75, 25, 105)
    assert r["bound"] == "fp64_valu" and r["achieved"] is None and r["hbm"]["achieved_GBs"] > 0
    latest, bench._latest = bench._latest, None
    try:  # ... and every older file of the same kinds, not only the latest
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_stage_profile.txt"))):
            bench._latest = lambda pattern, f=f: f
            s, _ = bench.contact_solve_share()
            assert s is None or 0.02 < s < 0.6, f
    finally:
        bench._latest = latest
