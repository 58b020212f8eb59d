"""Agent / AgentPG / AgentPPO with the reference's structure (uhc/khrylib/rl/agents/{agent,agent_pg,agent_ppo}.py)
re-designed around a batched on-device environment:

* ``Agent.sample`` steps all n_env environments together; states, actions, rewards, masks and exps are written
  into pre-allocated rollout tensors in HBM laid out [n_env][T] (each env's samples contiguous in time, the
  same order as the reference's concatenated per-worker memories).
* ``AgentPG.update_params`` runs value prediction, GAE (segment-parallel reverse scan), and the PPO epochs on
  those tensors without a host round trip.
* With ``torch.distributed`` initialised, every optimisation epoch all-reduces one flat buffer holding the
  policy and value gradients (summed over ranks, then divided by the global sample counts) before clipping,
  and the advantage statistics / observation-filter statistics are merged across ranks.
"""
from __future__ import annotations

import math
import os
import time
import types

import numpy as np
import torch
import torch.distributed as dist

from uhc_amd import rollout_ops
from uhc_amd import sim as _S  # field ids of the library's views

from ..core import LoggerRL, PolicyGaussian, estimate_advantages, linear
from ...utils.torch import to_test, to_train


def _dist_on():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class RolloutBatch:
    """What TrajBatch (uhc/khrylib/rl/core/trajbatch.py:5-15) holds, as device tensors [n_env*T, .]."""

    def __init__(self, states, actions, rewards, masks, exps, seg_len):
        self.states, self.actions, self.rewards, self.masks, self.exps, self.seg_len = states, actions, rewards, masks, exps, seg_len


class Agent:
    def __init__(self, env, policy_net, value_net, dtype, device, gamma, data_loader, custom_reward=None, end_reward=True,
                 mean_action=False, render=False, running_state=None, num_threads=1):
        self.env, self.policy_net, self.value_net = env, policy_net, value_net
        self.dtype, self.device, self.gamma = dtype, device, gamma
        self.custom_reward, self.end_reward, self.mean_action = custom_reward, end_reward, mean_action
        self.running_state, self.render, self.num_threads = running_state, render, num_threads
        self.noise_rate = 1.0
        self.logger_cls = LoggerRL
        self.sample_modules = [policy_net]
        self.update_modules = [policy_net, value_net]
        self.data_loader = data_loader

    def set_noise_rate(self, noise_rate):
        self.noise_rate = noise_rate

    def trans_policy(self, states):
        return states

    def trans_value(self, states):
        return states

    # ---- hooks the concrete agent provides -------------------------------------------------------------
    def assign_new_clips(self, env_ids):
        """Pick a clip window for each env in env_ids (host side: data_loader.sample_seq) and reset them."""
        raise NotImplementedError

    def queue_next_clips(self, env_ids):
        """Pick the window each env in env_ids starts after its current episode and hand it to env.set_next."""
        raise NotImplementedError

    def on_episode_end(self, env_ids, percents, consumed=None):
        pass

    # ---- vectorised rollout (replaces sample_worker / sample of agent.py:42-131) -------------------------
    # One control step = three pieces: `_seg_pre` (buffer write, policy forward, sampling -- torch ops), the library's env.step (PD target,
    # fused physics, termination, reward, observation: its own HIP kernels) and `_seg_post` (reward / done bookkeeping, observation filter,
    # device-side restart of finished episodes, next filtered state).  The torch pieces are ~60 small launches whose Python dispatch cost
    # more than their GPU time; on a GPU they are captured once into two HIP graphs (per pass length T) and replayed, the step index living
    # in a device counter.  Every buffer a graph touches is allocated once and updated in place across passes.
    use_graph = os.environ.get("UHC_NO_GRAPHS") != "1"  # class default; agent.use_graph = False (or UHC_NO_GRAPHS=1) runs the eager path (CPU runs always do)

    def _buffers(self, T):
        env = self.env
        n_env, dev = env.n_env, env.device
        cache = self.__dict__.setdefault("_ro_cache", {})
        R = cache.get(T)
        if R is None:
            R = cache[T] = types.SimpleNamespace(T=T)
            R.states = torch.empty(n_env, T, env.obs_dim, dtype=self.dtype, device=dev)
            R.actions = torch.empty(n_env, T, env.action_dim, dtype=self.dtype, device=dev)
            R.rewards = torch.zeros(n_env, T, dtype=self.dtype, device=dev)
            R.dones = torch.zeros(n_env, T, dtype=self.dtype, device=dev)
            R.mean_flags = torch.zeros(T, n_env, dtype=torch.float64, device=dev)
            R.c_info_sum = torch.zeros(getattr(env, "n_reward_parts", 5), dtype=self.dtype, device=dev)
            R.c_reward_sum = torch.zeros((), dtype=self.dtype, device=dev)
            R.t_dev = torch.zeros(1, dtype=torch.long, device=dev)
            R.redo_counts = torch.zeros(7, dtype=torch.long, device=dev)  # env-steps of this pass: through the general / large tier, its sweeps fallback, rows dropped beyond the last tier's capacity, large tier (or tier 4), exact solve in windows of 64 rows, solved by Newton on the primal (tier 4), Newton stopped at its iteration cap
            R.end_reward_dev = torch.zeros((), dtype=self.dtype, device=dev)
            R.state = torch.zeros(n_env, env.obs_dim, dtype=self.dtype, device=dev)
            R.action = torch.zeros(n_env, env.action_dim, dtype=torch.float64, device=dev)
            R.snap_host = [torch.empty(5, n_env, dtype=torch.float64).pin_memory() if dev.type == "cuda" else torch.empty(5, n_env, dtype=torch.float64) for _ in range(2)]
            R.graphs = None
        return R

    @torch.no_grad()
    def rollout_begin(self, T, fresh=None):
        """fresh: True = assign + reset every env (new episodes), False = continue the episodes left open by the previous pass,
        None = fresh on the first pass only."""
        env = self.env
        n_env, dev = env.n_env, env.device
        R = self._ro = self._buffers(T)
        R.t = 0
        R.t_dev.zero_(); R.redo_counts.zero_(); R.rewards.zero_(); R.dones.zero_(); R.c_info_sum.zero_(); R.c_reward_sum.zero_()
        R.end_reward_dev.fill_(float(env.end_reward) if self.end_reward else 0.0)
        R.logger = self.logger_cls()
        # exploration flags of the whole pass in one upload (the same stream of draws as one binomial(n_env) per step)
        flags = np.ones((T, n_env)) if self.mean_action else env.np_random.binomial(1, 1 - self.noise_rate, size=(T, n_env))
        R.mean_flags.copy_(torch.from_numpy(flags.astype(np.float64)))
        # episode turnover is asynchronous: the device restarts finished envs from a queued window (env.auto_reset); the
        # host learns about finished episodes from a pinned snapshot one step later and refills the queues then
        R.snap_event = [None, None]
        to_test(*self.sample_modules)
        if fresh is None:
            fresh = not getattr(self, "_episodes_live", False)
        if fresh:
            # first pass (or an explicit restart): every env starts a new episode.  Later passes CONTINUE the running episodes: the
            # reference's workers run every episode to `fail` or `end` (agent.py:60-100), so an episode cut by the end of a pass is
            # bootstrapped with V(s_T) in rollout_end and simply goes on here -- its state, running length / return, clip window and
            # queued successor all live on (device and agent bookkeeping), so env_episode_len, t_max and the success history of long
            # windows behave as in the reference.
            self.assign_new_clips(np.arange(n_env))
            self.queue_next_clips(np.arange(n_env))
            obs = env.obs.to(self.dtype)
            R.state.copy_(self.running_state(obs) if self.running_state is not None else obs)
            self._episodes_live = True
        else:
            # the observations the envs stopped at were already counted by the filter at the end of the previous pass; only
            # re-normalise them with the current statistics (the filter may have been merged across ranks since)
            obs = env.obs.to(self.dtype)
            R.state.copy_(self.running_state(obs, update=False) if self.running_state is not None else obs)

    def _drain_snapshot(self, slot):
        """Host side of episode turnover for the step whose snapshot sits in `slot`: statistics, success history, new queue entries."""
        R = self._ro
        ev = R.snap_event[slot]
        if ev is None:
            return
        if ev is not True:
            ev.synchronize()
        R.snap_event[slot] = None
        host = R.snap_host[slot].numpy()
        ids = np.nonzero(host[0])[0]
        if len(ids):
            R.logger.add_episodes(host[1][ids], host[2][ids])
            self.on_episode_end(ids, host[3][ids], consumed=host[4][ids] != 0)
            need = ids[host[4][ids] != 0]  # envs that took their queued window need a new one
            if len(need):
                self.queue_next_clips(need)

    def _seg_pre(self):
        """state -> rollout buffer, policy forward + sampling -> action (buffer + the fixed tensor env.step reads)."""
        R = self._ro
        t = R.t_dev
        pol = self.policy_net
        if isinstance(pol, PolicyGaussian) and rollout_ops.usable(R.state, R.states, R.actions, R.action, pol.action_log_std.data):
            # device float64: the trunk's GEMMs stay with the framework, everything around them is one launch of the library
            mean = linear(pol.action_mean, pol.net(self.trans_policy(R.state)))
            noise = torch.randn(mean.shape, dtype=mean.dtype, device=mean.device)  # the draw DiagGaussian.sample makes (same Philox stream)
            rollout_ops.act(t, R.state, mean.contiguous(), pol.action_log_std.data, noise, R.mean_flags, R.states, R.actions, R.action)
            return
        R.states.index_copy_(1, t, R.state.unsqueeze(1))
        mean_flag = R.mean_flags.index_select(0, t).squeeze(0)
        action = self.policy_net.select_action(self.trans_policy(R.state), mean_flag)
        R.actions.index_copy_(1, t, action.to(self.dtype).unsqueeze(1))
        R.action.copy_(action)

    def _seg_post(self):
        """reward / done bookkeeping, observation filter (the finished episodes' last observation included, agent.py:77-79), device-side
        restart of finished episodes, filtered state of the next step, step counter."""
        env, R = self.env, self._ro
        t = R.t_dev
        env.sim.use_current_stream()  # the library's launches below go to the stream this runs (or is being captured) on
        if rollout_ops.usable(env.reward, env.obs, R.rewards, R.dones, R.state, R.c_info_sum) and env.done.dtype == torch.int32:
            # device float64: six launches of the library (+ the restart's) instead of ~50 framework ones
            rollout_ops.record(t, env.reward, env.done, env.env.field(_S.E_END), R.end_reward_dev, env.env.field(_S.E_REWARD_PARTS), int(R.c_info_sum.numel()), R.rewards, R.dones,
                               R.c_reward_sum, R.c_info_sum, env.sim.field(_S.F_REDO), R.redo_counts)
            if self.running_state is not None:
                self.running_state.rs.push_batch(env.obs, weights=env.done)  # the finished episodes' last observations (agent.py:77-79)
            env.auto_reset()
            if self.running_state is not None:
                self.running_state(env.obs, out=R.state, step_counter=t)
            else:
                R.state.copy_(env.obs)
                t.add_(1)
            return
        redo = env.sim.field(_S.F_REDO)  # UHC_F_REDO: which envs the general kernel computed / solved by sweeps in this step (diagnostics)
        R.redo_counts[0] += ((redo & 1) != 0).sum()
        R.redo_counts[1] += ((redo & 2) != 0).sum()
        R.redo_counts[2] += ((redo & 0x80) != 0).sum()
        R.redo_counts[3] += ((redo & 0x40) != 0).sum()
        R.redo_counts[4] += ((redo & 8) != 0).sum()
        R.redo_counts[5] += ((redo & (1 << 30)) != 0).sum()
        R.redo_counts[6] += ((redo & (1 << 29)) != 0).sum()
        r = env.reward.to(self.dtype)
        R.c_reward_sum.add_(r.sum())  # the plain imitation reward: what LoggerRL reports (logger_rl.py:29-33), before end bonus / bootstrap
        if self.running_state is not None:
            self.running_state.rs.push_batch(env.obs.to(self.dtype), weights=env.done)  # the restart below overwrites those rows
        r = r + env.env.field(_S.E_END).to(self.dtype) * R.end_reward_dev  # info["end"] * end_reward (agent.py:84-85); 0 when switched off
        R.rewards.index_copy_(1, t, r.unsqueeze(1))
        R.dones.index_copy_(1, t, env.done.to(self.dtype).unsqueeze(1))  # masks = 1 - dones and exps = 1 - mean_flags are formed at the end of the pass
        R.c_info_sum.add_(env.reward_parts.sum(0))
        env.auto_reset()  # also writes the step's snapshot: done, episode length / return (kept by the library), percent, consumed
        obs = env.obs.to(self.dtype)
        R.state.copy_(self.running_state(obs) if self.running_state is not None else obs)
        t.add_(1)

    def _capture(self):
        """Capture the two torch segments into HIP graphs (torch.cuda.CUDAGraph = hipGraph on ROCm).  The capture records the work
        without running it; the caller replays."""
        env, R = self.env, self._ro
        torch.cuda.synchronize()
        g_pre, g_post = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # No finaliser may run while a stream is capturing: the cyclic garbage of an EARLIER agent / env (a test suite makes several per process) ends in
        # uhc_batch_free / uhc_env_free -- hipFree, event and stream destruction -- which a capture in global mode does not allow.  Collect what is pending first,
        # then hold the collector back until both captures are over (uhc_amd/sim.py defers such a free as well, should one come from elsewhere).
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(g_pre):
                self._seg_pre()
            with torch.cuda.graph(g_post, pool=g_pre.pool()):
                self._seg_post()
        finally:
            if gc_was_on:
                gc.enable()
        env.sim.use_current_stream()  # back from the capture stream
        R.graphs = (g_pre, g_post)
        R.graph_gen = getattr(getattr(env, "env", None), "generation", 0)

    @torch.no_grad()
    def rollout_step(self):
        """One control step of every env: filter -> policy -> env.step -> buffers -> device-side restart of finished episodes.
        The host reads the previous step's pinned snapshot while the GPU computes this one, then waits for this step (see below)."""
        env, R = self.env, self._ro
        if R.t >= R.T:
            raise RuntimeError(f"rollout_step: the pass was begun for {R.T} steps")
        dev = env.device
        graphed = self.use_graph and dev.type == "cuda"
        if graphed and R.graphs is not None and R.graph_gen != getattr(getattr(env, "env", None), "generation", 0):
            R.graphs = None  # the clip bank / clip models were replaced: the captured env launches carry the old device pointers by value
        if graphed and R.graphs is None and R.t >= 2:  # two eager steps first: library handles, allocator pools and the filter's device state exist
            self._capture()
        if graphed and R.graphs is not None:
            R.graphs[0].replay()
            env.step(R.action)
            R.graphs[1].replay()
            if self.running_state is not None:
                self.running_state.rs.mark_device_updated()
        else:
            self._seg_pre()
            env.step(R.action)
            self._seg_post()
        slot = R.t & 1
        self._drain_snapshot(slot)  # the snapshot written two steps ago (its buffer is reused now)
        R.snap_host[slot].copy_(env.env.field(_S.E_SNAPSHOT), non_blocking=True)
        if dev.type == "cuda":
            R.snap_event[slot] = torch.cuda.Event()
            R.snap_event[slot].record()
        else:
            R.snap_event[slot] = True
        # the previous step's snapshot landed before that step returned: its host work (episode statistics, sampling the next windows,
        # ~0.8 ms at 1 024 envs) runs while the GPU computes this step
        self._drain_snapshot(slot ^ 1)
        # ... and the host then waits for this step, so the GPU queue never holds more than one control step.  Letting the host run a
        # step ahead measured slower and erratic (4.1 - 8 ms per step against a steady 4.15): see DESIGN.md section 6.
        if dev.type == "cuda":
            R.snap_event[slot].synchronize()
        R.t += 1

    @torch.no_grad()
    def rollout_end(self):
        env, R = self.env, self._ro
        T, N = R.T, env.n_env * R.T
        self._drain_snapshot((T - 1) & 1)  # the last step's episode ends
        R.masks = 1 - R.dones
        R.exps = (1 - R.mean_flags).t().contiguous().to(self.dtype)
        rewards = R.rewards.clone()  # the pass's buffers are reused by the next pass: the batch owns what it changes
        # episodes cut by the end of the pass: bootstrap with V(s_T) folded into the last reward, then close the segment
        open_ = R.masks[:, T - 1] > 0
        if bool(open_.any()):
            v_next = self.value_net(self.trans_value(R.state[open_])).squeeze(-1)
            rewards[open_, T - 1] += self.gamma * v_next
            R.masks[open_, T - 1] = 0
        self.sync_running_state()
        R.logger.add_steps(N, float(R.c_reward_sum.item()), R.c_info_sum.cpu().numpy())
        R.logger.end_sampling()
        # Lifetime: `states` / `actions` / `masks` / `exps` are VIEWS of this pass length's buffers, which the next rollout_begin(T) of the
        # same T overwrites in place (the captured graphs keep their addresses); a caller that keeps a batch across passes clones it.
        return RolloutBatch(R.states.reshape(N, -1), R.actions.reshape(N, -1), rewards.reshape(N, 1), R.masks.reshape(N, 1), R.exps.reshape(N), T), R.logger

    def sync_running_state(self):
        """Data-parallel runs: merge the observation-filter statistics of all ranks (Chan), so every rank
        normalises with the same mean/std and rank 0's checkpoint holds the global filter."""
        if self.running_state is None or not _dist_on():
            return
        rs = self.running_state.rs
        dev = self.env.device if self.env is not None else torch.device("cpu")
        if dist.get_backend() != "nccl":
            dev = torch.device("cpu")  # gloo gathers host tensors only
        mine = torch.cat([torch.tensor([float(rs.n)], dtype=torch.float64), torch.from_numpy(np.asarray(rs.mean, dtype=np.float64).ravel()),
                          torch.from_numpy(np.asarray(rs._S, dtype=np.float64).ravel())]).to(dev)
        allv = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allv, mine)
        d = (mine.numel() - 1) // 2
        base = getattr(self, "_rs_synced", None)  # statistics every rank already shares (previous merge, or a loaded checkpoint)
        n0, M0, S0 = (0.0, np.zeros(d), np.zeros(d)) if base is None else base
        n, M, Sq = n0, M0.copy(), S0.copy()
        for v in allv:  # add each rank's NEW samples: (rank stats) minus (shared base), merged pairwise
            v = v.cpu().numpy()
            nr, Mr, Sr = v[0], v[1:1 + d], v[1 + d:]
            nb = nr - n0
            if nb <= 0:
                continue
            mb = (nr * Mr - n0 * M0) / nb
            delta0 = mb - M0
            Sb = Sr - S0 - delta0 * delta0 * (n0 * nb / nr)
            tot = n + nb
            delta = mb - M
            Sq = Sq + Sb + delta * delta * (n * nb / tot)
            M = M + delta * (nb / tot)
            n = tot
        self.running_state.set_mean_std(M.reshape(rs._M.shape), Sq.reshape(rs._S.shape), int(round(n)))
        self._rs_synced = (float(rs._n), M.copy(), Sq.copy())

    def mark_running_state_shared(self):
        """The filter statistics are the same on every rank right now (loaded from one checkpoint): later merges add only what each
        rank observes from here on."""
        if self.running_state is not None:
            rs = self.running_state.rs
            self._rs_synced = (float(rs.n), np.asarray(rs.mean, dtype=np.float64).ravel().copy(), np.asarray(rs._S, dtype=np.float64).ravel().copy())

    def sample(self, min_batch_size):
        t0 = time.time()
        T = int(math.ceil(min_batch_size / self.env.n_env))
        self.rollout_begin(T)
        for _ in range(T):
            self.rollout_step()
        batch, logger = self.rollout_end()
        logger.sample_time = time.time() - t0
        return batch, logger


class AgentPG(Agent):
    def __init__(self, tau=0.95, optimizer_policy=None, optimizer_value=None, opt_num_epochs=1, value_opt_niter=1, **kwargs):
        super().__init__(**kwargs)
        self.tau, self.optimizer_policy, self.optimizer_value = tau, optimizer_policy, optimizer_value
        self.opt_num_epochs, self.value_opt_niter = opt_num_epochs, value_opt_niter
        self._flat_grad = None

    # ---- data-parallel gradient exchange: ONE collective per optimisation step ----------------------------
    def _allreduce_grads(self, params_and_scale):
        """params_and_scale: list of (parameter list, local sample count) -- gradients are local means; turn them
        into global means: sum_r (n_r * g_r) / sum_r n_r, with one all-reduce over a flat buffer."""
        if not _dist_on():
            return
        self._finish_grad_exchange(self._start_grad_exchange(params_and_scale, slot=0))

    def _start_grad_exchange(self, params_and_scale, slot=0):
        """Pack the local gradient means (times the rank's sample count) and the counts into flat buffer `slot` and START the all-reduce
        (`async_op`: over RCCL it runs on the process group's own stream, so whatever the caller enqueues next -- the surrogate's backward
        pass -- overlaps with it).  Returns what `_finish_grad_exchange` needs."""
        plist = [p for ps, _ in params_and_scale for p in ps if p.grad is not None]
        total = sum(p.numel() for p in plist) + len(params_and_scale)
        # the wire: float32 by default (`grad_allreduce_dtype`, SURVEY 8e's 32 MB per epoch), or the parameters' own dtype (float64 like
        # the reference's arithmetic: the 2-rank update then equals the single-process one to rounding).  The local gradient MEANS go on
        # the wire scaled by the rank's sample count: |n g| stays far inside float32.
        wire = getattr(self, "grad_wire_dtype", None) or plist[0].dtype
        if self._flat_grad is None:
            self._flat_grad = {}
        cached = self._flat_grad.get(slot)
        key = tuple(id(p) for p in plist)
        if cached is None or cached[0].numel() != total or cached[0].dtype != wire or cached[2] != key:
            # the wire buffer and, once, the parameter-shaped views into it: packing and unpacking are then a handful of multi-tensor launches
            # (`torch._foreach_*`: one horizontally fused kernel per group) instead of two small launches per parameter -- 16 per exchange, 20
            # exchanges per update, which an 8-GPU step that SURVEY 8e calls latency-bound would see
            buf = torch.empty(total, dtype=wire, device=plist[0].device)
            views, off = [], 0
            for ps, _ in params_and_scale:
                group = []
                for p in ps:
                    if p.grad is None:
                        continue
                    group.append(buf[off:off + p.numel()].view_as(p))
                    off += p.numel()
                views.append(group)
            cached = self._flat_grad[slot] = (buf, views, key)
        buf, views, _ = cached
        off = total - len(params_and_scale)
        for (ps, n), group in zip(params_and_scale, views):
            grads = [p.grad for p in ps if p.grad is not None]
            if grads:
                torch._foreach_copy_(group, torch._foreach_mul(grads, float(n)))  # (the copy casts to the wire's dtype)
        buf[off:].copy_(torch.tensor([float(n) for _, n in params_and_scale], dtype=wire), non_blocking=True)  # (sample counts up to 2^24 per rank are exact in float32)
        ev = None
        if getattr(self, "time_comm", False) and buf.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        return work, buf, off, params_and_scale, ev

    def _finish_grad_exchange(self, started):
        work, buf, off, params_and_scale, ev = started
        if ev is not None:
            mid = torch.cuda.Event(enable_timing=True)
            mid.record()
        work.wait()
        if ev is not None:
            ev[1].record()
            self._comm_events = getattr(self, "_comm_events", [])
            self._comm_events.append((ev[0], ev[1], buf.numel() * buf.element_size(), mid))
        counts = buf[off:off + len(params_and_scale)]
        views = next(v for b, v, _ in self._flat_grad.values() if b is buf)
        for i, ((ps, _), group) in enumerate(zip(params_and_scale, views)):
            grads = [p.grad for p in ps if p.grad is not None]
            if grads:
                torch._foreach_copy_(grads, group)  # (back to the parameters' dtype first: the division is the reference's float64 one)
                torch._foreach_div_(grads, counts[i].to(grads[0].dtype).clamp(min=1.0))

    def comm_summary(self):
        """(calls, total ms, mean bytes per call) of the gradient all-reduces timed since the last summary (time_comm = True).  An interval runs
        from the start of an exchange to the end of the wait for it: with `overlap_grad_exchange` it CONTAINS the backward pass enqueued in
        between; `self.comm_exposed_ms` is the part the compute stream actually stood still for (wait issued -> wait over)."""
        ev = getattr(self, "_comm_events", [])
        self._comm_events = []
        self.comm_exposed_ms = 0.0
        if not ev:
            return 0, 0.0, 0
        torch.cuda.synchronize()
        self.comm_exposed_ms = float(sum(m.elapsed_time(b) for _, b, _, m in ev))
        return len(ev), float(sum(a.elapsed_time(b) for a, b, _, _ in ev)), int(sum(e[2] for e in ev) // len(ev))

    def update_value(self, states, returns):
        for _ in range(self.value_opt_niter):
            value_loss = (self.value_net(self.trans_value(states)) - returns).pow(2).mean()
            self.optimizer_value.zero_grad()
            value_loss.backward()
            self._allreduce_grads([(list(self.value_net.parameters()), states.shape[0])])
            self.optimizer_value.step()
        return value_loss.detach()

    def update_policy(self, states, actions, returns, advantages, exps):
        ind = exps.nonzero().squeeze(1)
        for _ in range(self.opt_num_epochs):
            self.update_value(states, returns)
            log_probs = self.policy_net.get_log_prob(self.trans_policy(states)[ind], actions[ind])
            policy_loss = -(log_probs * advantages[ind]).mean()
            self.optimizer_policy.zero_grad()
            policy_loss.backward()
            self._allreduce_grads([(list(self.policy_net.parameters()), ind.shape[0])])
            self.optimizer_policy.step()

    def update_params(self, batch):
        t0 = time.time()
        to_train(*self.update_modules)
        states, actions, rewards, masks, exps = batch.states, batch.actions, batch.rewards, batch.masks, batch.exps
        with to_test(*self.update_modules), torch.no_grad():
            values = self.value_net(self.trans_value(states))
        reduce = None
        if _dist_on():
            def reduce(v):
                dist.all_reduce(v, op=dist.ReduceOp.SUM)
                return v
        advantages, returns = estimate_advantages(rewards, masks, values, self.gamma, self.tau, seg_len=getattr(batch, "seg_len", None),
                                                  stats_reduce=reduce)
        self.update_policy(states, actions, returns, advantages, exps)
        return time.time() - t0


class AgentPPO(AgentPG):
    def __init__(self, clip_epsilon=0.2, mini_batch_size=64, use_mini_batch=False, policy_grad_clip=None, **kwargs):
        super().__init__(**kwargs)
        self.clip_epsilon, self.mini_batch_size, self.use_mini_batch = clip_epsilon, mini_batch_size, use_mini_batch
        self.policy_grad_clip = policy_grad_clip

    def update_policy(self, states, actions, returns, advantages, exps):
        """agent_ppo.py:16-51: the full-batch branch (every copycat config) and the mini-batch one (`use_mini_batch`, :23-43)."""
        with to_test(*self.update_modules), torch.no_grad():
            fixed_log_probs = self.policy_net.get_log_prob(self.trans_policy(states), actions)
        if self.use_mini_batch:
            return self._update_policy_mini_batch(states, actions, returns, advantages, fixed_log_probs, exps)
        ind = exps.nonzero(as_tuple=False).squeeze(1)
        self.last_losses = []
        fused = _dist_on() and self.value_opt_niter == 1 and getattr(self, "fuse_grad_exchange", True)
        for _ in range(self.opt_num_epochs):
            if fused:
                # data-parallel: ONE exchange per optimisation epoch: the value gradient and the surrogate's gradient travel in
                # one flat buffer.  The surrogate does not read the value net (the advantages are fixed for the update), so taking both
                # gradients before either Adam step is the same computation as the reference's value step followed by the policy step.
                value_loss = (self.value_net(self.trans_value(states)) - returns).pow(2).mean()
                surr_loss = self.ppo_loss(states, actions, advantages, fixed_log_probs, ind)
                self.optimizer_value.zero_grad()
                self.optimizer_policy.zero_grad()
                value_loss.backward()
                if getattr(self, "overlap_grad_exchange", True):
                    # the value gradient is ready first: its half of the exchange travels while the surrogate's backward pass runs
                    # (two collectives of half the size each per epoch; the same sums as the single buffer)
                    first = self._start_grad_exchange([(list(self.value_net.parameters()), states.shape[0])], slot=1)
                    surr_loss.backward()
                    second = self._start_grad_exchange([([p for p in self.policy_net.parameters() if p.requires_grad], ind.shape[0])], slot=2)
                    self._finish_grad_exchange(first)
                    self._finish_grad_exchange(second)
                else:
                    surr_loss.backward()
                    self._allreduce_grads([(list(self.value_net.parameters()), states.shape[0]),
                                           ([p for p in self.policy_net.parameters() if p.requires_grad], ind.shape[0])])
                self.optimizer_value.step()
                self.clip_policy_grad()
                self.optimizer_policy.step()
                self.last_losses.append((value_loss.detach(), surr_loss.detach()))
                continue
            vl = self.update_value(states, returns)
            surr_loss = self.ppo_loss(states, actions, advantages, fixed_log_probs, ind)
            self.optimizer_policy.zero_grad()
            surr_loss.backward()
            self._allreduce_grads([([p for p in self.policy_net.parameters() if p.requires_grad], ind.shape[0])])
            self.clip_policy_grad()
            self.optimizer_policy.step()
            self.last_losses.append((vl, surr_loss.detach()))

    def _update_policy_mini_batch(self, states, actions, returns, advantages, fixed_log_probs, exps):
        """agent_ppo.py:23-43.  Per optimisation epoch: a permutation of the batch drawn from numpy's GLOBAL generator (`np.random.shuffle`) and applied to the
        ALREADY permuted arrays (the reference reassigns its arguments: the permutations compose over the epochs), then floor(N / mini_batch_size) steps --
        the ragged tail is dropped -- each a value step on the mini-batch followed by a surrogate step on its `exps != 0` rows.  Data-parallel: every
        rank permutes its own rows, the ranks take the same number of steps (the fewest any rank has) and exchange gradients in every one of them."""
        self.last_losses = []
        mbs = self.mini_batch_size
        for _ in range(self.opt_num_epochs):
            perm = np.arange(states.shape[0])
            np.random.shuffle(perm)
            perm = torch.as_tensor(perm, dtype=torch.long, device=states.device)
            states, actions, returns, advantages, fixed_log_probs, exps = (v[perm].clone() for v in (states, actions, returns, advantages, fixed_log_probs, exps))
            n_it = int(math.floor(states.shape[0] / mbs))
            if _dist_on():
                t = torch.tensor([n_it], device=states.device)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                n_it = int(t.item())
            for i in range(n_it):
                sl = slice(i * mbs, min((i + 1) * mbs, states.shape[0]))
                states_b, actions_b, advantages_b, returns_b, fixed_b, exps_b = states[sl], actions[sl], advantages[sl], returns[sl], fixed_log_probs[sl], exps[sl]
                ind = exps_b.nonzero(as_tuple=False).squeeze(1)
                vl = self.update_value(states_b, returns_b)
                surr_loss = self.ppo_loss(states_b, actions_b, advantages_b, fixed_b, ind)
                self.optimizer_policy.zero_grad()
                surr_loss.backward()
                self._allreduce_grads([([p for p in self.policy_net.parameters() if p.requires_grad], ind.shape[0])])
                self.clip_policy_grad()
                self.optimizer_policy.step()
                self.last_losses.append((vl, surr_loss.detach()))

    def clip_policy_grad(self):
        if self.policy_grad_clip is not None:
            for params, max_norm in self.policy_grad_clip:
                params = list(params)  # the reference passes a generator: exhausted after the first call, every later call clips nothing
                if params:
                    torch.nn.utils.clip_grad_norm_(params, max_norm)

    def ppo_loss(self, states, actions, advantages, fixed_log_probs, ind):
        log_probs = self.policy_net.get_log_prob(self.trans_policy(states)[ind], actions[ind])
        ratio = torch.exp(log_probs - fixed_log_probs[ind])
        adv = advantages[ind]
        surr1 = ratio * adv
        surr2 = torch.clamp(ratio, 1.0 - self.clip_epsilon, 1.0 + self.clip_epsilon) * adv
        return -torch.min(surr1, surr2).mean()
