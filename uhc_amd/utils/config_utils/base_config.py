"""yml -> attribute bag (mirror of uhc/utils/config_utils/base_config.py:9-62).

`Config(cfg_id)` finds exactly one `config/**/<cfg_id>.yml` under `base_dir` (the reference's layout),
falling back to the yml files shipped with this package (config/uhc_amd/)."""
import glob
import os
import os.path as osp
import shutil

import yaml

PKG_ROOT = osp.dirname(osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))


def recreate_dirs(*dirs):
    for d in dirs:
        if osp.exists(d):
            shutil.rmtree(d)
        os.makedirs(d)


class Base_Config:
    def __init__(self, cfg_id, base_dir="", create_dirs=False, cfg_dict=None):
        self.id = cfg_id
        self.base_dir = osp.expanduser(base_dir if base_dir else "")
        if cfg_dict is None:
            files = glob.glob(osp.join(self.base_dir, f"config/**/{cfg_id}.yml"), recursive=True)
            if not files:
                files = glob.glob(osp.join(PKG_ROOT, f"config/**/{cfg_id}.yml"), recursive=True)
            if len(files) != 1:
                raise FileNotFoundError(f"expected exactly one config/**/{cfg_id}.yml, found {len(files)}")
            with open(files[0]) as f:
                cfg_dict = yaml.safe_load(f)
        cfg = self.cfg_dict = cfg_dict
        self.main_result_dir = osp.join(self.base_dir, "results")
        self.proj_name = cfg.get("proj_name", "motion_im")
        self.cfg_dir = osp.join(self.main_result_dir, self.proj_name, cfg_id)
        self.model_dir = osp.join(self.cfg_dir, "models")
        self.output_dir = self.result_dir = osp.join(self.cfg_dir, "results")
        self.log_dir = osp.join(self.cfg_dir, "log")
        os.makedirs(self.model_dir, exist_ok=True)
        os.makedirs(self.output_dir, exist_ok=True)
        if create_dirs and not osp.exists(self.log_dir):
            recreate_dirs(self.log_dir)
        os.makedirs(self.log_dir, exist_ok=True)
        self.seed = cfg.get("seed", 1)
        self.notes = cfg.get("notes", "exp notes")
        self.data_specs = cfg.get("data_specs", {})
        self.loss_specs = cfg.get("loss_specs", {})
        self.model_specs = cfg.get("model_specs", {})
        self.lr = cfg.get("lr", 3.0e-4)
        self.num_epoch = cfg.get("num_epoch", 100)
        self.num_epoch_fix = cfg.get("num_epoch_fix", 10)
        self.save_n_epochs = cfg.get("save_n_epochs", 20)
        self.eval_n_epochs = cfg.get("eval_n_epochs", 20)
        self.num_samples = self.data_specs.get("num_samples", 5000)
        self.batch_size = self.data_specs.get("batch_size", 5000)

    def get(self, key, default=None):
        return self.cfg_dict.get(key, default)

    def update(self, args):
        for k, v in vars(args).items():
            setattr(self, k, v)
