"""Train the Universal Humanoid Controller (copycat) policy -- CLI of the reference's scripts/train_uhc.py:34-99.

    python scripts/train_uhc.py --cfg copycat_mi355x [--epoch N] [--gpu_index 0] [--no_log] [--synthetic 64]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_uhc.py --cfg copycat_mi355x

One process per GPU; with torch.distributed initialised the envs are sharded across ranks and the PPO gradients
are all-reduced over RCCL every optimisation epoch (uhc_amd/khrylib/rl/agents).  wandb is optional."""
import argparse
import os
import os.path as osp
import sys

sys.path.append(os.getcwd())
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))

import numpy as np
import torch

from uhc.agents import agent_dict  # the reference's import paths (scripts/train_uhc.py:30-32), served by uhc/__init__.py -> uhc_amd
from uhc.utils.config_utils.copycat_config import Config
from uhc.utils.flags import flags

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--cfg", default=None)
    parser.add_argument("--render", action="store_true", default=False)
    parser.add_argument("--test", action="store_true", default=False)
    parser.add_argument("--num_threads", type=int, default=30)
    parser.add_argument("--gpu_index", type=int, default=0)
    parser.add_argument("--epoch", type=int, default=0)
    parser.add_argument("--show_noise", action="store_true", default=False)
    parser.add_argument("--resume", type=str, default=None)
    parser.add_argument("--no_log", action="store_true", default=False)
    parser.add_argument("--debug", action="store_true", default=False)
    parser.add_argument("--full_eval", action="store_true", default=False)
    parser.add_argument("--synthetic", type=int, default=0, help="train on N synthetic clips instead of data_specs.file_path")
    parser.add_argument("--num_epoch", type=int, default=None)
    parser.add_argument("--n_env", type=int, default=None, help="batched environments per GPU (default: the config's; 1 = the reference's single-env plumbing case)")
    parser.add_argument("--min_batch_size", type=int, default=None, help="samples per iteration (default: the config's)")
    args = parser.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        args.gpu_index = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # every rank evaluates its share of the test clips every save_n_epochs (eval_policy deals the keys out over the ranks by index, not
        # by clip length), so on full AMASS with few ranks one rank may lag the others by a long evaluation before the next collective: the
        # timeout is generous (UHC_DIST_TIMEOUT_HOURS, default 6) -- a hung rank still ends the job, a slow one does not
        import datetime
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo",
                                timeout=datetime.timedelta(hours=float(os.environ.get("UHC_DIST_TIMEOUT_HOURS", "6"))))
    cfg = Config(cfg_id=args.cfg, create_dirs=not (args.render or args.epoch > 0))
    over = {k: getattr(args, k) for k in ("num_epoch", "n_env", "min_batch_size")}
    for k in over:
        delattr(args, k)
    cfg.update(args)
    for k, v in over.items():
        if v is not None:
            setattr(cfg, k, v)
    flags.debug = args.debug
    if not cfg.no_log:
        try:
            import wandb
            wandb.init(project="copycat", resume=args.resume is not None, id=args.resume, notes=cfg.notes)
        except ImportError:
            cfg.no_log = True
    dtype = torch.float64
    torch.set_default_dtype(dtype)
    if not torch.cuda.is_available():
        raise SystemExit("uhc_amd needs an MI355X: the batched environment has no CPU fallback")
    device = torch.device("cuda", index=args.gpu_index)
    torch.cuda.set_device(args.gpu_index)
    rank = int(os.environ.get("RANK", "0"))
    np.random.seed(cfg.seed + rank)
    torch.manual_seed(cfg.seed)  # identical initial weights on every rank
    data_loader = None
    if args.synthetic:
        from uhc_amd.data_loaders.dataset_amass_single import DatasetAMASSSingle
        from uhc_amd.data_loaders.synthetic import make_synthetic_amass
        specs = dict(cfg.data_specs)
        specs["file_path"] = "synthetic"
        data_loader = DatasetAMASSSingle(specs, "train", pickle_data=make_synthetic_amass(args.synthetic, seed=1 + rank))
    agent = agent_dict[cfg.agent_name](cfg, dtype, device, training=True, checkpoint_epoch=args.epoch, data_loader=data_loader)
    for i_iter in range(args.epoch, cfg.num_epoch):
        agent.optimize_policy(i_iter)
    print("training done!")
