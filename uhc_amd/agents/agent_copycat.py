"""AgentCopycat (mirror of uhc/agents/agent_copycat.py:53-605): same constructor, ``optimize_policy``,
``sample``, ``update_params``, ``save_checkpoint`` / ``load_checkpoint`` (same file names and dict keys),
driving the batched MI355X environment instead of forked CPU workers."""
from __future__ import annotations

import logging
import os
import os.path as osp
import pickle
import time

import numpy as np
import torch
import torch.distributed as dist

from ..data_loaders.dataset_amass_single import DatasetAMASSSingle
from ..envs.humanoid_im import VecHumanoidEnv
from ..khrylib.models.mlp import MLP
from ..khrylib.rl.agents import AgentPPO, _dist_on
from ..khrylib.rl.core import PolicyGaussian, Value
from ..khrylib.utils.logger import create_logger
from ..khrylib.utils.torch import get_eta_str, lambda_rule, set_optimizer_lr, to_device
from ..khrylib.utils.zfilter import ZFilter
from ..losses.reward_function import DEVICE_REWARD_IDS


def cfg_get(cfg, key, default):
    return getattr(cfg, key, default)


class AgentCopycat(AgentPPO):
    def __init__(self, cfg, dtype, device, training=True, checkpoint_epoch=0, data_loader=None, shape_models=None, clip_model=None, body_provider=None,
                 objects=None):
        """shape_models / clip_model: optional body shapes (models with the topology of the config's model) and the map clip key -> index
        into [config model] + shape_models: smpl_shape-style training where every clip runs on its own body.
        body_provider: callable (betas, gender) -> (vertices, joints, skin weights) feeding the shape -> model generator (default: the SMPL
        model files under <base_dir>/data/smpl when present): every clip then runs on the model generated from ITS beta and gender, as
        the reference rebuilds its model at every load_expert (uhc/envs/humanoid_im.py:154-190).
        objects: free objects behind every env's humanoid (VecHumanoidEnv(objects=): the meshes the reference's generator reads from
        expert["obj_info"], uhc/smpllib/smpl_robot.py:1200-1252); every clip of the data set then carries `obj_pose` (T, 7 K)."""
        self.cfg = self.cc_cfg = cfg
        self._shape_models, self._clip_model, self._body_provider, self._objects = shape_models, clip_model, body_provider, objects
        self.device, self.dtype, self.training = device, dtype, training
        self.max_freq = 50
        self.epoch = 0
        self.precision_mode = cfg.get("precision_mode", False)
        self.fit_single_key = ""  # scripts/fit_uhc.py: sample every window from this one clip (agent_copycat.py:102, :504-510)
        self.setup_data_loader(data_loader)
        self.setup_env()
        self.setup_policy()
        self.setup_value()
        self.setup_optimizer()
        self.setup_logging()
        self.setup_reward()
        self.seed(cfg.seed)
        if checkpoint_epoch > 0:
            self.load_checkpoint(checkpoint_epoch)
            self.epoch = checkpoint_epoch
            self._loaded_shared_filter = True
        super().__init__(env=self.env, dtype=dtype, device=device, running_state=self.running_state, custom_reward=self.expert_reward,
                         mean_action=bool(getattr(cfg, "render", False)) and not getattr(cfg, "show_noise", False), render=False,
                         num_threads=getattr(cfg, "num_threads", 1), data_loader=self.data_loader, policy_net=self.policy_net,
                         value_net=self.value_net, optimizer_policy=self.optimizer_policy, optimizer_value=self.optimizer_value,
                         opt_num_epochs=cfg.num_optim_epoch, gamma=cfg.gamma, tau=cfg.tau, clip_epsilon=cfg.clip_epsilon,
                         policy_grad_clip=[(self.policy_net.parameters(), 40)], end_reward=cfg.end_reward, use_mini_batch=False,
                         mini_batch_size=0)
        self.grad_wire_dtype = {"float32": torch.float32, "float64": torch.float64}[str(getattr(cfg, "grad_allreduce_dtype", "float32"))]
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1 and torch.distributed.get_rank() == 0:
            # (the wire defaults to float32 since round 5 -- an unchanged multi-rank yml is then within 1e-4 relative of the single-process update, not 1e-10)
            print(f"[uhc_amd] gradient exchange over {torch.distributed.get_world_size()} ranks: wire dtype {cfg_get(cfg, 'grad_allreduce_dtype', 'float32')}, "
                  f"overlap_grad_exchange {cfg_get(cfg, 'overlap_grad_exchange', True)} (set grad_allreduce_dtype: float64 for parity / regression runs)", flush=True)
        self.overlap_grad_exchange = bool(getattr(cfg, "overlap_grad_exchange", True))
        if getattr(self, "_loaded_shared_filter", False):
            self.mark_running_state_shared()  # every rank loaded the same filter statistics: they are not new samples

    # ---- setup ------------------------------------------------------------------------------------------
    def setup_data_loader(self, data_loader=None):
        self.data_loader = data_loader if data_loader is not None else DatasetAMASSSingle(self.cfg.data_specs, data_mode="train")
        self.test_data_loaders = [self.data_loader]
        if len(self.cfg.data_specs.get("test_file_path", [])) > 0 and data_loader is None:
            self.test_data_loaders.append(DatasetAMASSSingle(self.cfg.data_specs, data_mode="test"))

    def setup_env(self):
        dev_index = self.device.index if isinstance(self.device, torch.device) and self.device.index is not None else 0
        model = None
        if self._shape_models is None:
            from ..smpllib.smpl_robot import default_body_provider, generate_shape_models
            provider = self._body_provider or default_body_provider(self.cfg)
            if provider is not None:  # reset_robot for every distinct (beta, gender) of the data set, once
                dl = self.data_loader
                clips = {k: dict(beta=dl.data["beta"][k], gender=dl.data["gender"][k]) for k in dl.data_keys}
                for tl in self.test_data_loaders[1:]:
                    clips.update({k: dict(beta=tl.data["beta"][k], gender=tl.data["gender"][k]) for k in tl.data_keys})
                models, self._clip_model = generate_shape_models(self.cfg.robot_cfg, clips, provider)
                model, self._shape_models = models[0], models[1:]
        self.env = VecHumanoidEnv(self.cfg, n_env=self.cfg.n_env, device=dev_index, mode="train", model=model, shape_models=self._shape_models, objects=self._objects)
        self.env.set_clip_bank_from_loader(self.data_loader, clip_model={k: v for k, v in self._clip_model.items() if k in set(self.data_loader.data_keys)} if self._clip_model else None)

    def setup_policy(self):
        cfg, env = self.cfg, self.env
        self.state_dim, self.action_dim = env.observation_space.shape[0], env.action_space.shape[0]
        if cfg.actor_type == "gauss":
            self.policy_net = PolicyGaussian(cfg, action_dim=self.action_dim, state_dim=self.state_dim)
        elif cfg.actor_type == "mcp":
            from ..models.policy_mcp import PolicyMCP
            self.policy_net = PolicyMCP(cfg, action_dim=self.action_dim, state_dim=self.state_dim)
        else:
            raise ValueError(f"actor_type {cfg.actor_type!r}")
        self.running_state = ZFilter((self.state_dim,), clip=5)
        to_device(self.device, self.policy_net)

    def setup_value(self):
        self.value_net = Value(MLP(self.state_dim, self.cfg.value_hsize, self.cfg.value_htype))
        to_device(self.device, self.value_net)

    def setup_optimizer(self):
        cfg = self.cfg

        def make(name, params, lr, mom, wd):
            if name == "Adam":
                return torch.optim.Adam(params, lr=lr, weight_decay=wd)
            return torch.optim.SGD(params, lr=lr, momentum=mom, weight_decay=wd)

        self.optimizer_policy = make(cfg.policy_optimizer, self.policy_net.parameters(), cfg.policy_lr, cfg.policy_momentum, cfg.policy_weightdecay)
        self.optimizer_value = make(cfg.value_optimizer, self.value_net.parameters(), cfg.value_lr, cfg.value_momentum, cfg.value_weightdecay)

    def setup_reward(self):
        if self.cfg.reward_id not in DEVICE_REWARD_IDS:
            raise NotImplementedError(f"reward '{self.cfg.reward_id}' is a later row (SURVEY.md 8f-4)")
        self.expert_reward = self.cfg.reward_id

    def setup_logging(self):
        cfg = self.cfg
        freq_path = osp.join(cfg.result_dir, "freq_dict.pt")
        self.freq_dict = {k: [] for k in self.data_loader.data_keys}
        if osp.exists(freq_path):
            try:
                import joblib
                fd = joblib.load(freq_path)
                if set(fd.keys()) == set(self.data_loader.data_keys):
                    self.freq_dict = fd
            except Exception:
                pass
        import torch.distributed as dist
        rank = dist.get_rank() if _dist_on() else 0
        self.logger = create_logger(os.path.join(cfg.log_dir, "log.txt" if rank == 0 else f"log_rank{rank}.txt"))

    def seed(self, seed):
        """Data-parallel ranks share the weights (initialised before this call from the script's global seed, and broadcast from
        rank 0 below) but must draw different clip windows, exploration flags and action noise: their sampling streams are seeded
        with seed + rank (rank 0 alone reproduces the single-process run)."""
        import torch.distributed as dist
        self.rank = dist.get_rank() if _dist_on() else 0
        if _dist_on():
            for p in list(self.policy_net.parameters()) + list(self.value_net.parameters()):
                dist.broadcast(p.data, 0)
        torch.manual_seed(seed + self.rank)
        np.random.seed(seed + self.rank)
        self.env.seed(seed + self.rank)

    # ---- checkpoints: file names, dict keys and CPU float64 state_dict layout as agent_copycat.py:190-276 ----
    def _cp(self):
        return {"policy_dict": {k: v.detach().cpu() for k, v in self.policy_net.state_dict().items()},
                "value_dict": {k: v.detach().cpu() for k, v in self.value_net.state_dict().items()}, "running_state": self.running_state}

    def save_checkpoint(self, epoch):
        cfg = self.cfg
        if getattr(self, "rank", 0) != 0:  # one writer: the weights are identical on every rank, the filter is merged (sync_running_state)
            return
        pickle.dump(self._cp(), open("%s/iter_%04d.p" % (cfg.model_dir, epoch + 1), "wb"))
        try:
            import joblib
            joblib.dump(self.freq_dict, osp.join(cfg.result_dir, "freq_dict.pt"))
        except ImportError:
            pass

    def save_curr(self):
        if getattr(self, "rank", 0) != 0:
            return
        pickle.dump(self._cp(), open(f"{self.cfg.model_dir}/iter_best.p", "wb"))

    def save_singles(self, epoch, key):  # agent_copycat.py:203-214
        if getattr(self, "rank", 0) != 0:
            return
        os.makedirs(f"{self.cfg.model_dir}_singles", exist_ok=True)
        pickle.dump(self._cp(), open(f"{self.cfg.model_dir}_singles/{key}.p", "wb"))

    def load_singles(self, epoch, key):  # agent_copycat.py:238-247
        if epoch > 0:
            self._load(f"{self.cfg.model_dir}/iter_{(epoch + 1):04d}_{key}.p")

    def _load(self, path):
        self.logger.info("loading model from checkpoint: %s" % path)
        cp = CustomUnpickler(open(path, "rb")).load()
        self.policy_net.load_state_dict(cp["policy_dict"])
        self.value_net.load_state_dict(cp["value_dict"])
        rs = cp["running_state"]  # copied INTO the agent's filter: its device tensors are referenced by the captured rollout graphs
        if rs is not None and self.running_state is not None:  # (the reference accepts checkpoints without a filter: the current one stays)
            self.running_state.set_mean_std(np.asarray(rs.rs.mean), np.asarray(rs.rs._S), int(rs.rs.n))
            self.running_state.demean, self.running_state.destd, self.running_state.clip = rs.demean, rs.destd, rs.clip
        to_device(self.device, self.policy_net, self.value_net)
        # every rank has just loaded the same statistics: they are the shared base of the next merge, not new samples of this rank
        # (only once the Agent base exists -- __init__ loads before it and marks the base itself)
        if hasattr(self, "mark_running_state_shared") and hasattr(self, "env") and getattr(self, "_ro_cache", None) is not None:
            self.mark_running_state_shared()

    def load_curr(self):
        self._load(f"{self.cfg.model_dir}/iter_best.p")

    def load_checkpoint(self, iter):
        if iter > 0:
            self._load("%s/iter_%04d.p" % (self.cfg.model_dir, iter))

    # ---- clip sampling hooks used by the vectorised rollout ----------------------------------------------------
    def _sample_windows(self, n):
        cfg, dl = self.cfg, self.data_loader
        if self.fit_single_key != "":  # single-clip fitting: every window comes from one clip (agent_copycat.py:504-510)
            keys, fs, fe = [], [], []
            for _ in range(n):
                dl.get_sample_from_key(self.fit_single_key, full_sample=False, freq_dict=self.freq_dict, precision_mode=self.precision_mode)
                keys.append(dl.curr_key)
                fs.append(dl.fr_start)
                fe.append(dl.fr_end)
            return keys, fs, fe
        if self.precision_mode:  # per-draw path of the reference (window near a recorded failure)
            keys, fs, fe = [], [], []
            for _ in range(n):
                dl.sample_seq(freq_dict=self.freq_dict, full_sample=False, sampling_temp=cfg.sampling_temp, sampling_freq=cfg.sampling_freq,
                              precision_mode=True)
                keys.append(dl.curr_key)
                fs.append(dl.fr_start)
                fe.append(dl.fr_end)
            return keys, fs, fe
        return dl.sample_windows(n, freq_dict=self.freq_dict, sampling_temp=cfg.sampling_temp, sampling_freq=cfg.sampling_freq)

    def assign_new_clips(self, env_ids):
        """load_expert + reset on the listed envs right now (start of a sampling pass)."""
        keys, fs, fe = self._sample_windows(len(env_ids))
        self._env_key = getattr(self, "_env_key", {})
        for e, k, s in zip(env_ids, keys, fs):
            self._env_key[int(e)] = (k, int(s))
        self.env.assign(np.asarray(env_ids), keys, fs, fe)
        self.env.reset(np.asarray(env_ids))

    def queue_next_clips(self, env_ids):
        """Sample the window each listed env starts when its current episode ends; the device does the restart itself."""
        keys, fs, fe = self._sample_windows(len(env_ids))
        self._env_next = getattr(self, "_env_next", {})
        for e, k, s in zip(env_ids, keys, fs):
            self._env_next[int(e)] = (k, int(s))
        self.env.set_next(np.asarray(env_ids), keys, fs, fe)

    def on_episode_end(self, env_ids, percents, consumed=None):
        for n, (e, p) in enumerate(zip(env_ids, percents)):  # freq_dict[key].append([percent, fr_start]) (agent_copycat.py:559-565)
            k, s = self._env_key[int(e)]
            self._freq_add(k, [[float(p), s]])
            if consumed is not None and consumed[n]:  # the env has moved on to its queued window
                self._env_key[int(e)] = self._env_next[int(e)]

    # ---- training iteration ------------------------------------------------------------------------------
    def per_epoch_update(self, epoch):
        cfg = self.cfg
        cfg.update_adaptive_params(epoch)
        self.set_noise_rate(cfg.adp_noise_rate)
        set_optimizer_lr(self.optimizer_policy, cfg.adp_policy_lr)
        if cfg.rfc_decay:
            mx = cfg.get("rfc_decay_max", 10000)
            self.env.set_rfc_rate(lambda_rule(self.epoch, mx, cfg.num_epoch_fix) if self.epoch < mx else 0.0)
        if cfg.fix_std:
            self.policy_net.action_log_std.data.fill_(cfg.adp_log_std)

    def optimize_policy(self, epoch, save_model=True):
        cfg = self.cfg
        self.epoch = epoch
        t0 = time.time()
        self.per_epoch_update(epoch)
        batch, log = self.sample(cfg.min_batch_size)
        if cfg.end_reward:
            # one value on every rank (the sample-weighted mean over the ranks' logs), or the ranks' reward shaping drifts apart
            self.env.end_reward = self._global_mean(log.avg_c_reward, log.num_steps) * cfg.gamma / (1 - cfg.gamma)
        t1 = time.time()
        self.update_params(batch)
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        t2 = time.time()
        info = {"log": log, "T_sample": t1 - t0, "T_update": t2 - t1, "T_total": t2 - t0}
        if save_model and (self.epoch + 1) % cfg.save_n_epochs == 0:
            self.save_checkpoint(epoch)  # (rank 0 writes)
            info["log_eval"] = self.eval_policy(epoch)  # every rank evaluates its share of the clips (keys[rank::world]); the metrics are reduced
        self.sync_freq_dict()  # every iteration, as the reference merges its workers' histories after every sample(); eval-driven entries included
        if getattr(self, "rank", 0) == 0:
            self.log_train(info)
        return info

    def _global_mean(self, value, weight):
        if not _dist_on():
            return float(value)
        dev = self.device if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([float(value) * float(weight), float(weight)], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        return float(t[0].item() / max(t[1].item(), 1e-300))

    def _freq_add(self, key, rows):
        """freq_dict[key] += rows, remembered as this rank's contribution since the last merge (sync_freq_dict)."""
        self.freq_dict[key] = (self.freq_dict[key] + rows)[-self.max_freq:]
        new = self.__dict__.setdefault("_freq_new", {})
        new[key] = (new.get(key, []) + rows)[-self.max_freq:]

    def sync_freq_dict(self):
        """Data-parallel runs: merge the per-clip success history of all ranks, as the reference merges its workers' (agent_copycat.py:
        598-604: concatenate per key, keep the last `max_freq`), the evaluation entries included -- every rank then draws its next
        windows from the same distribution.  Each rank contributes what it appended since the last merge; the history before that is
        already common.  On the wire: flat (clip index, value, first frame) float64 records, one count exchange + one padded all-gather
        (tensors over RCCL / gloo, no pickling of Python objects on the host)."""
        if not _dist_on():
            return
        new = self.__dict__.setdefault("_freq_new", {})
        keys = list(self.freq_dict.keys())  # the data set's clip keys: the same order on every rank
        dev = self.device if dist.get_backend() == "nccl" else torch.device("cpu")
        rows = [[float(ki), float(r[0]), float(r[1])] for ki, k in enumerate(keys) for r in new.get(k, [])]
        world = dist.get_world_size()
        cnt = torch.tensor([len(rows)], dtype=torch.int64, device=dev)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        cnts = [int(c.item()) for c in cnts]
        add = {k: [] for k in keys}
        if max(cnts) > 0:
            mine = torch.zeros(max(cnts), 3, dtype=torch.float64, device=dev)
            if rows:
                mine[:len(rows)] = torch.tensor(rows, dtype=torch.float64, device=dev)
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            for p, c in zip(parts, cnts):  # rank order: the same list on every rank
                for ki, v, fs in p[:c].cpu().tolist():
                    add[keys[int(ki)]].append([v, int(fs)])
        for k in keys:
            mine_n = len(new.get(k, []))
            base = self.freq_dict[k][:len(self.freq_dict[k]) - mine_n] if mine_n else self.freq_dict[k]
            self.freq_dict[k] = (base + add[k])[-self.max_freq:]
        self._freq_new = {}

    def log_train(self, info):
        log, cfg = info["log"], self.cfg
        c = np.array2string(log.avg_c_info, formatter={"all": lambda x: "%.4f" % x}, separator=",")
        self.logger.info(f"Ep: {self.epoch}\t {cfg.id} \tT_s {info['T_sample']:.2f}\t T_u {info['T_update']:.2f}\tETA "
                         f"{get_eta_str(self.epoch, cfg.num_epoch, info['T_total'])}\texpert_R_avg {log.avg_c_reward:.4f} {c}"
                         f"\texpert_R_range ({log.min_c_reward:.4f}, {log.max_c_reward:.4f})\teps_len {log.avg_episode_len:.2f}")


# ---- evaluation (agent_copycat.py:354-494), batched: one env per test clip --------------------------------
def _eval_policy(self, epoch=0, dump=False):
    """agent_copycat.py:354-436.  Data-parallel: the clips of every loader are dealt out over the ranks (keys[rank::world]); the per-clip
    metrics are summed over the ranks (one small all-reduce per loader) and every rank logs / returns the same global means; the
    clip-success entries go into each rank's freq_dict contribution and reach the others with the next sync_freq_dict."""
    from collections import defaultdict
    cfg = self.cfg
    res_dicts = []
    names = ["mpjpe", "mpjpe_g", "accel_dist", "vel_dist", "succ", "reward", "root_dist", "pentration", "skate"]
    world, rank = (dist.get_world_size(), dist.get_rank()) if _dist_on() else (1, 0)
    for loader in self.test_data_loaders:
        keys = list(loader.data_keys)[rank::world]
        cov = self.eval_seqs(keys, loader) if keys else {}
        for k, res in cov.items():
            if k in self.freq_dict:
                self._freq_add(k, [[res["succ"][0], 0]] * (1 if res["succ"][0] else 3))
        m = defaultdict(list)
        for res in cov.values():
            for k, v in res.items():
                if k in names:
                    m[k].append(v if np.ndim(v) == 0 else np.mean(v))
        present = sorted(m.keys())
        if world > 1:  # sums and counts of every metric over the ranks
            dev = self.device if dist.get_backend() == "nccl" else torch.device("cpu")
            t = torch.tensor([[float(np.sum(m.get(k, []))), float(len(m.get(k, [])))] for k in names], dtype=torch.float64, device=dev)
            dist.all_reduce(t)
            m = {k: float(t[i, 0] / t[i, 1]) for i, k in enumerate(names) if t[i, 1] > 0}
        else:
            m = {k: float(np.mean(m[k])) for k in present}
        coverage = int(m["succ"] * loader.get_len())
        if rank == 0:
            self.logger.info(f"Coverage {loader.name} of {coverage} out of {loader.get_len()} | " + " \t".join(f"{k}: {v:.3f}" for k, v in m.items()))
        m.update({"mean_coverage": coverage / loader.get_len(), "num_coverage": coverage, "all_coverage": loader.get_len()})
        del m["succ"]
        res_dicts.append({f"coverage_{loader.name}": m})
        if dump:
            import joblib
            if world > 1:
                parts = [None] * world
                dist.all_gather_object(parts, cov)  # (the per-clip trajectories: evaluation dumps only)
                cov = {k: v for p in parts for k, v in p.items()}
            if rank == 0:
                joblib.dump(cov, osp.join(cfg.output_dir, f"{epoch}_{loader.name}_coverage_full.pkl"))
    return res_dicts


@torch.no_grad()
def _eval_seqs(self, take_keys, loader):
    """Run the mean-action policy over whole clips (eval_seq, agent_copycat.py:438-494), all clips of a chunk at once."""
    from ..smpllib.smpl_eval import compute_metrics
    from .. import sim as S
    cfg = self.cfg
    take_keys = list(take_keys)
    n = min(len(take_keys), cfg.n_env)
    ev = getattr(self, "_eval_envs", {}).get((loader.name, n))
    if ev is None:
        ev = VecHumanoidEnv(cfg, n_env=n, device=self.env.device.index or 0, mode="test", model=self.env.body_model, shape_models=self.env.body_models[1:],
                            objects=self.env.objects)
        cm = getattr(self, "_clip_model", None)  # every clip is evaluated on its own body, as it is trained (a clip the map does not name: body 0)
        ev.set_clip_bank_from_loader(loader, clip_model={k: cm.get(k, 0) for k in loader.data_keys} if cm else None)
        self._eval_envs = getattr(self, "_eval_envs", {})
        self._eval_envs[(loader.name, n)] = ev
    ev.set_rfc_rate(self.env.rfc_rate)
    out = {}
    frames, starts = ev.env._bank[0], ev.env._bank[1].cpu().numpy()
    for c0 in range(0, len(take_keys), n):
        keys = take_keys[c0:c0 + n]
        m = len(keys)
        ids = np.arange(m)
        lens = np.array([loader.get_sample_len_from_key(k) for k in keys])
        ev.assign(ids, keys, np.zeros(m, dtype=int), lens)
        ev.reset(ids)
        # the reference filters the reset observation with update=True (agent_copycat.py:445-446); kept
        state = self.running_state(ev.obs[:m].to(self.dtype)) if self.running_state is not None else ev.obs[:m].to(self.dtype)
        if m < n:
            state = torch.cat([state, state.new_zeros(n - m, state.shape[1])])
        active = torch.zeros(n, dtype=torch.int32, device=ev.device)
        active[:m] = 1
        rec = {k: [[] for _ in range(m)] for k in ("gt", "pred", "gt_jpos", "pred_jpos", "reward")}
        fail_safe = np.zeros(m, dtype=bool)
        alive = np.ones(m, dtype=bool)
        clip0 = np.array([starts[ev._clip_index[k]] for k in keys])
        t = np.zeros(m, dtype=int)

        ql, bl = ev.qpos_lim, ev.body_lim  # the humanoid's part of qpos / of the bodies (objects come behind it)

        def snap():
            q = ev.sim.field(S.F_QPOS)[:m, :ql].cpu().numpy()
            x = ev.sim.field(S.F_XPOS)[:m].cpu().numpy()[:, 3:3 * bl]
            gi = clip0 + np.minimum(t, lens - 1)
            g = frames[torch.from_numpy(gi).to(frames.device)].cpu().numpy()
            for e in np.nonzero(alive)[0]:
                rec["pred"][e].append(q[e]); rec["pred_jpos"][e].append(x[e])
                rec["gt"][e].append(g[e, 0:76]); rec["gt_jpos"][e].append(g[e, 151:223])
            return g

        percent = np.zeros(m)
        while alive.any():
            snap()
            action = self.policy_net.select_action(self.trans_policy(state), True).to(torch.float64).contiguous()
            ev.step(action, active)
            done = ev.done[:m].cpu().numpy().astype(bool) & alive
            r = ev.reward[:m].cpu().numpy()
            pct = ev.env.field(S.E_PERCENT)[:m].cpu().numpy()
            for e in np.nonzero(alive)[0]:
                rec["reward"][e].append(r[e])
            if done.any():
                q = ev.sim.field(S.F_QPOS)[:m, :ql].cpu().numpy()
                x = ev.sim.field(S.F_XPOS)[:m].cpu().numpy()[:, 3:3 * bl]
                gi = clip0 + np.minimum(t, lens - 1)
                g = frames[torch.from_numpy(gi).to(frames.device)].cpu().numpy()
                tele = []
                for e in np.nonzero(done)[0]:
                    rec["pred"][e].append(q[e]); rec["pred_jpos"][e].append(x[e])
                    rec["gt"][e].append(g[e, 0:76]); rec["gt_jpos"][e].append(g[e, 151:223])
                    if cfg.fail_safe and pct[e] != 1:  # teleport to the expert state and keep going (humanoid_im.py:902-905)
                        fail_safe[e] = True
                        tele.append(e)
                    else:
                        alive[e] = False
                        active[e] = 0
                        percent[e] = pct[e]
                if tele:
                    tele = np.array(tele)
                    cur = ev.cur_t[:m].cpu().numpy()[tele]
                    fr = frames[torch.from_numpy(clip0[tele] + np.minimum(cur, lens[tele] - 1)).to(frames.device)]
                    # the expert pose in the model's own coordinates: hinge angles, or -- ball joints -- root pose + the joints' quaternions
                    qp, qv = S.expert_pose_of_frames(fr, ev.use_quat)
                    if ev.num_obj:  # data.qpos[:qpos_lim] = expert pose (humanoid_im.py:902-905): the objects stay where they are
                        ti = torch.from_numpy(tele).to(frames.device)
                        qp = torch.cat([qp, ev.sim.field(S.F_QPOS)[ti, ql:]], 1)
                        qv = torch.cat([qv, ev.sim.field(S.F_QVEL)[ti, ev.qvel_lim:]], 1)
                    ev.sim.set_state(qp.contiguous(), qv.contiguous(), torch.from_numpy(tele).to(torch.int32))
            t = t + 1
            state = self.running_state(ev.obs.to(self.dtype), update=False) if self.running_state is not None else ev.obs.to(self.dtype)
        for e, k in enumerate(keys):
            res = {kk: np.vstack(v[e]) for kk, v in rec.items() if kk != "reward"}
            res["reward"] = np.array(rec["reward"][e])
            res["percent"], res["fail_safe"] = percent[e], bool(fail_safe[e])
            res.update(compute_metrics(res, None))
            out[k] = res
    return out


from ..utils.tools import CustomUnpickler  # noqa: E402,F401  (the reference keeps it in uhc/utils/tools.py; re-exported here for older callers)


def _eval_seq(self, take_key, loader):
    """One clip, first frame to last, mean action (agent_copycat.py:438-494): the result dict of that clip."""
    return self.eval_seqs([take_key], loader)[take_key]


AgentCopycat.eval_policy = _eval_policy
AgentCopycat.eval_seqs = _eval_seqs
AgentCopycat.eval_seq = _eval_seq
