"""Fit the policy to the clips it cannot track yet, one clip at a time -- the loop of the reference's scripts/fit_uhc.py:96-134.

    python scripts/fit_uhc.py --cfg <id> [--iter N] [--synthetic 16] [--max_epochs M]

Starting from `models/iter_best.p`: evaluate the current clip with the mean action; while it fails, run PPO iterations
whose every window is drawn from that clip (`agent.fit_single_key`, precision mode) and overwrite `iter_best.p`; once it is
tracked to its last frame store the weights as `models_singles/<clip>.p`, reload `iter_best.p` and go on to the next clip
(clips already present in `models_singles/` are skipped)."""
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


def fit(agent, start_epoch=0, max_epochs=99999, log=print):
    """The per-clip loop; returns {clip: epoch at which it was fitted}."""
    cfg = agent.cfg
    os.makedirs(f"{cfg.model_dir}_singles", exist_ok=True)
    done_keys = [osp.splitext(k)[0] for k in os.listdir(f"{cfg.model_dir}_singles/")]
    take_keys = iter([k for k in agent.data_loader.data_keys if k not in done_keys])
    fitted = {}
    try:
        take_key = next(take_keys)
    except StopIteration:
        return fitted
    for epoch in range(start_epoch, max_epochs):
        res = agent.eval_seq(take_key, agent.data_loader)
        if not np.all(res["succ"]):
            log(f"Fitting: {take_key} {bool(np.all(res['succ']))}")
            agent.fit_single_key = take_key
            agent.optimize_policy(epoch, save_model=False)
            agent.save_curr()
        else:
            log(f"************************Fitted {take_key} at {epoch}")
            agent.save_singles(epoch, take_key)
            fitted[take_key] = epoch
            try:
                take_key = next(take_keys)
            except StopIteration:
                break
            agent.load_curr()
    agent.fit_single_key = ""
    return fitted


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--cfg", default=None)
    parser.add_argument("--num_threads", type=int, default=40)
    parser.add_argument("--gpu_index", type=int, default=0)
    parser.add_argument("--iter", type=int, default=0)
    parser.add_argument("--no_log", action="store_false", default=True)
    parser.add_argument("--debug", action="store_true", default=False)
    parser.add_argument("--synthetic", type=int, default=0, help="fit N synthetic clips instead of data_specs.file_path")
    parser.add_argument("--max_epochs", type=int, default=99999)
    args = parser.parse_args()

    cfg = Config(cfg_id=args.cfg, create_dirs=not args.iter > 0)
    cfg.update(args)
    flags.debug = args.debug
    cfg.no_log = True
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
    agent = AgentCopycat(cfg, dtype, device, training=True, checkpoint_epoch=args.iter, data_loader=data_loader)
    agent.precision_mode = True
    if not osp.exists(f"{cfg.model_dir}/iter_best.p"):
        agent.save_curr()  # nothing trained yet: start from the current (random or --iter) weights
    agent.load_curr()
    fit(agent, args.iter, args.max_epochs)
