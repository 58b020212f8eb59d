"""Repeat the PPO update on one collected batch: run-to-run spread of the learner time on one box."""
import os, sys, time, tempfile
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
torch.set_default_dtype(torch.float64)
from uhc_amd.agents.agent_copycat import AgentCopycat
from uhc_amd.data_loaders.dataset_amass_single import DatasetAMASSSingle
from uhc_amd.data_loaders.synthetic import make_synthetic_amass
from uhc_amd.utils.config_utils.copycat_config import Config
cfg = Config(cfg_id="copycat_mi355x", base_dir=tempfile.mkdtemp())
cfg.no_log = True
specs = dict(cfg.data_specs); specs["file_path"] = "synthetic"
dl = DatasetAMASSSingle(specs, "train", pickle_data=make_synthetic_amass(64, seed=1))
agent = AgentCopycat(cfg, torch.float64, torch.device("cuda", 0), data_loader=dl)
agent.per_epoch_update(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 60
agent.rollout_begin(T)
for _ in range(T): agent.rollout_step()
batch, _ = agent.rollout_end()
for r in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agent.update_params(batch)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"update {r}: {el:.3f} s  {batch.states.shape[0] / el:.0f} samples/s")
