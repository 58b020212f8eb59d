"""GPU diagnostic: wall time of a rollout step with / without HIP graphs and with / without the library's bookkeeping kernels,
and of the two graph replays alone."""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
torch.set_default_dtype(torch.float64)
from uhc_amd import rollout_ops
from uhc_amd.agents.agent_copycat import AgentCopycat
from uhc_amd.data_loaders.dataset_amass_single import DatasetAMASSSingle
from uhc_amd.data_loaders.synthetic import make_synthetic_amass
from uhc_amd.utils.config_utils.copycat_config import Config

usable0 = rollout_ops.usable
combos = [(bool(int(sys.argv[1])), bool(int(sys.argv[2])))] if len(sys.argv) > 2 else [(g, f) for g in (True, False) for f in (True, False)]
for graph, fused in combos:
    if True:
        rollout_ops.usable = usable0 if fused else (lambda *a: False)
        cfg = Config(cfg_id="copycat_mi355x", base_dir=tempfile.mkdtemp(prefix="uhc_diag_"))
        cfg.n_env, cfg.no_log = 1024, True
        specs = dict(cfg.data_specs); specs["file_path"] = "synthetic"
        torch.manual_seed(1); np.random.seed(1)
        dl = DatasetAMASSSingle(specs, "train", pickle_data=make_synthetic_amass(64, seed=1))
        agent = AgentCopycat(cfg, torch.float64, torch.device("cuda", 0), data_loader=dl)
        agent.use_graph = graph
        agent.per_epoch_update(0)
        T = 100  # 10 warm-up + 40 timed steps + 2 x 20 stand-alone graph replays
        agent.rollout_begin(T)
        for _ in range(10):
            agent.rollout_step()
        acc = {}
        def wrap(obj, name, key):
            f = getattr(obj, name)
            def g(*a, **k):
                t = time.perf_counter(); r = f(*a, **k); acc[key] = acc.get(key, 0.0) + time.perf_counter() - t; return r
            setattr(obj, name, g)
        if os.environ.get("CPUPROF") == "1":
            wrap(agent, "_drain_snapshot", "drain"); wrap(agent.env, "step", "env.step"); wrap(agent, "on_episode_end", "on_episode_end"); wrap(agent, "queue_next_clips", "queue_next")
            wrap(agent, "rollout_step", "rollout_step")
        torch.cuda.synchronize()
        timing, redo = os.environ.get("TIMING") == "1", os.environ.get("REDO") == "1"
        from uhc_amd import sim as S
        if timing:
            agent.env.sim.set_timing(True)
        redo_acc = torch.zeros(1024, dtype=torch.int32, device="cuda")
        t0 = time.perf_counter()
        for _ in range(40):
            agent.rollout_step()
            if redo:
                redo_acc += (agent.env.sim.field(S.F_REDO) != 0).int()
            ex = os.environ.get("EXTRA", "")
            if ex == "a":
                redo_acc += 1
            elif ex == "b":
                tmp = agent.env.sim.field(S.F_REDO) != 0
            elif ex == "c":
                redo_acc.add_(agent.env.sim.field(S.F_REDO))
            elif ex == "d":
                tmp = redo_acc != 0
            elif ex == "e":
                tmp = (redo_acc != 0).int()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 40 * 1e3
        ex = os.environ.get("EXTRA", "")
        line = f"graph={graph} fused={fused} timing={timing} redo={redo} extra={ex}: {ms:.3f} ms/step"
        if acc:
            line += " | cpu ms/step: " + ", ".join(f"{k} {v / 40 * 1e3:.3f}" for k, v in acc.items())
        R = agent._ro
        if R.graphs is not None and not acc:
            for name, g in zip(("pre", "post"), R.graphs):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                e0.record()
                for _ in range(20):
                    g.replay()
                e1.record()
                cpu_ms = (time.perf_counter() - t1) / 20 * 1e3
                torch.cuda.synchronize()
                line += f" | {name}: gpu {e0.elapsed_time(e1) / 20:.3f} ms, cpu launch {cpu_ms:.3f} ms"
        print(line, flush=True)
        del agent
