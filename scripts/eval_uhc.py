"""Evaluate a trained UHC policy -- the `--mode stats` path of the reference's scripts/eval_uhc.py:36-103.

    python scripts/eval_uhc.py --cfg <id> --epoch N [--data <amass pickle>] [--no_fail_safe] [--synthetic 16]

Runs `AgentCopycat.eval_policy(epoch, dump=True)`: every clip of the data set is tracked from its first frame with the mean
action, all clips at once on the device; prints success rate, mpjpe, mpjpe_g, velocity / acceleration error and writes
`results/.../<epoch>_<name>_coverage_full.pkl`.  The reference's viewer modes (`vis`, `disp_stats`) need MuJoCo's renderer
and are outside this build."""
import argparse
import os
import os.path as osp
import sys

sys.path.append(os.getcwd())
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))

import numpy as np
import torch

from uhc_amd.utils.config_utils.copycat_config import Config
from uhc_amd.utils.flags import flags

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--cfg", default=None)
    parser.add_argument("--test", action="store_true", default=False)
    parser.add_argument("--num_threads", type=int, default=30)
    parser.add_argument("--gpu_index", type=int, default=0)
    parser.add_argument("--epoch", type=int, default=0)
    parser.add_argument("--no_log", action="store_true", default=False)
    parser.add_argument("--debug", action="store_true", default=False)
    parser.add_argument("--data", type=str, default="sample_data/amass_copycat_take5_test_small.pkl")
    parser.add_argument("--mode", type=str, default="stats")
    parser.add_argument("--no_fail_safe", action="store_true", default=False)
    parser.add_argument("--output", type=str, default="test")
    parser.add_argument("--synthetic", type=int, default=0, help="evaluate on N synthetic clips instead of --data")
    args = parser.parse_args()
    if args.mode != "stats":
        raise SystemExit("only --mode stats is built (the viewer modes need MuJoCo's renderer)")
    cfg = Config(cfg_id=args.cfg, create_dirs=False)
    cfg.update(args)
    flags.debug = args.debug
    cfg.no_log = True
    if args.no_fail_safe:
        cfg.fail_safe = False
    cfg.data_specs["file_path"] = args.data
    cfg.data_specs.pop("test_file_path", None)
    dtype = torch.float64
    torch.set_default_dtype(dtype)
    if not torch.cuda.is_available():
        raise SystemExit("uhc_amd needs an MI355X: the batched environment has no CPU fallback")
    torch.cuda.set_device(args.gpu_index)
    device = torch.device("cuda", index=args.gpu_index)
    np.random.seed(cfg.seed)
    torch.manual_seed(cfg.seed)
    data_loader = None
    if args.synthetic:
        from uhc_amd.data_loaders.dataset_amass_single import DatasetAMASSSingle
        from uhc_amd.data_loaders.synthetic import make_synthetic_amass
        specs = dict(cfg.data_specs)
        specs["file_path"] = "synthetic"
        data_loader = DatasetAMASSSingle(specs, "train", pickle_data=make_synthetic_amass(args.synthetic, seed=1))
    from uhc_amd.agents.agent_copycat import AgentCopycat
    agent = AgentCopycat(cfg, dtype, device, training=True, checkpoint_epoch=args.epoch, data_loader=data_loader)
    for res in agent.eval_policy(epoch=args.epoch, dump=True):
        print(res)
