"""RL building blocks with the reference's names (uhc/khrylib/rl/core/*):
estimate_advantages (GAE), DiagGaussian, Policy, PolicyGaussian, Value, LoggerRL."""
import math

import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Normal

from ...models.mlp import MLP, linear


def estimate_advantages(rewards, masks, values, gamma, tau, seg_len=None, normalize=True, stats_reduce=None):
    """Generalised advantage estimation (uhc/khrylib/rl/core/common.py:5-25) as torch ops on the device the
    rollout lives on.

    rewards, masks, values: (N, 1).  The recursion runs backwards over the flat batch exactly like the
    reference:  delta_i = r_i + gamma V_{i+1} m_i - V_i,  A_i = delta_i + gamma tau A_{i+1} m_i.
    With ``seg_len=T`` the batch is n_env contiguous segments of T steps whose LAST mask is 0 (episode end,
    or truncation with the bootstrap value already folded into the reward), so no chain crosses a segment and
    all segments are scanned in parallel: T vector steps instead of N scalar ones.
    returns = V + A;  A <- (A - mean) / std (unbiased), over the whole batch -- or over all ranks when
    ``stats_reduce`` (a callable summing a 3-vector [sum, sum of squares, count] across ranks) is given."""
    N = rewards.shape[0]
    if seg_len is None:
        seg_len = N
    assert N % seg_len == 0
    r, m, v = (x.reshape(N // seg_len, seg_len) for x in (rewards, masks, values))
    adv = torch.empty_like(r)
    prev_v = torch.zeros_like(r[:, 0])
    prev_a = torch.zeros_like(r[:, 0])
    for t in range(seg_len - 1, -1, -1):
        delta = r[:, t] + gamma * prev_v * m[:, t] - v[:, t]
        prev_a = delta + gamma * tau * prev_a * m[:, t]
        adv[:, t] = prev_a
        prev_v = v[:, t]
    advantages = adv.reshape(N, 1)
    returns = values + advantages
    if normalize:
        if stats_reduce is None:
            advantages = (advantages - advantages.mean()) / advantages.std()
        else:
            s = stats_reduce(torch.stack([advantages.sum(), (advantages ** 2).sum(), torch.tensor(float(N), dtype=advantages.dtype, device=advantages.device)]))
            mean = s[0] / s[2]
            var = (s[1] - s[2] * mean * mean) / (s[2] - 1)
            advantages = (advantages - mean) / torch.sqrt(var)
    return advantages, returns


class DiagGaussian(Normal):
    """distributions.py:6-25: log_prob summed over action dims, keepdim.  Built without argument validation: torch's check
    (`constraint.check(loc).all()` in a Python `if`) is a device-to-host sync on every policy call, i.e. once per rollout step."""

    def __init__(self, loc, scale):
        super().__init__(loc, scale, validate_args=False)

    def kl(self):
        loc1, scale1 = self.loc, self.scale
        loc0, scale0 = loc1.detach(), scale1.detach()
        kl = scale1.log() - scale0.log() + (scale0.pow(2) + (loc0 - loc1).pow(2)) / (2.0 * scale1.pow(2)) - 0.5
        return kl.sum(1, keepdim=True)

    def log_prob(self, value):
        return super().log_prob(value).sum(1, keepdim=True)

    def mean_sample(self):
        return self.loc

    def sample(self, sample_shape=torch.Size()):
        """loc + scale * N(0, 1): what Normal.sample draws (torch.normal(loc, scale) is randn * scale + loc on the same Philox stream),
        without torch.normal's host-side `std >= 0` check -- a device-to-host sync per call, and illegal while the rollout step is being
        captured into a HIP graph."""
        with torch.no_grad():
            shape = self._extended_shape(sample_shape)
            return self.loc.expand(shape) + self.scale.expand(shape) * torch.randn(shape, dtype=self.loc.dtype, device=self.loc.device)


class Policy(nn.Module):
    def select_action(self, x, mean_action=False):
        dist = self.forward(x)
        if torch.is_tensor(mean_action):  # batched env: per-row choice between the mean and a sample
            return torch.where(mean_action.reshape(-1, 1).bool(), dist.mean_sample(), dist.sample())
        return dist.mean_sample() if mean_action else dist.sample()

    def get_kl(self, x):
        return self.forward(x).kl()

    def get_log_prob(self, x, action):
        return self.forward(x).log_prob(action)


class PolicyGaussian(Policy):
    """policy_gaussian.py:9-31: MLP trunk -> Linear mean (x0.1 init, zero bias); state-independent log_std."""

    def __init__(self, cfg, action_dim, state_dim, net_out_dim=None):
        super().__init__()
        self.type = "gaussian"
        self.net = MLP(state_dim, cfg.policy_hsize, cfg.policy_htype)
        self.action_mean = nn.Linear(net_out_dim or self.net.out_dim, action_dim)
        self.action_mean.weight.data.mul_(0.1)
        self.action_mean.bias.data.mul_(0.0)
        self.action_log_std = nn.Parameter(torch.ones(1, action_dim) * cfg.log_std, requires_grad=not cfg.fix_std)

    def forward(self, x):
        mean = linear(self.action_mean, self.net(x))
        return DiagGaussian(mean, torch.exp(self.action_log_std.expand_as(mean)))


class Value(nn.Module):
    """critic.py:5-18: trunk -> Linear(., 1) (x0.1 init, zero bias)."""

    def __init__(self, net, net_out_dim=None):
        super().__init__()
        self.net = net
        self.value_head = nn.Linear(net_out_dim or net.out_dim, 1)
        self.value_head.weight.data.mul_(0.1)
        self.value_head.bias.data.mul_(0.0)

    def forward(self, x):
        return linear(self.value_head, self.net(x))


class TrajBatch:
    """The reference's host-side batch (uhc/khrylib/rl/core/trajbatch.py:5-15): the workers' `Memory` objects merged in list order, every field
    stacked -- states, actions, masks, next_states, rewards, exps, in the order `Memory.push` received them.  (This build's sampler fills
    `RolloutBatch` on the device instead; `TrajBatch` serves callers that still collect through `Memory`.)"""
    FIELDS = ("states", "actions", "masks", "next_states", "rewards", "exps")

    def __init__(self, memory_list):
        merged = memory_list[0]
        for m in memory_list[1:]:
            merged.append(m)
        columns = list(zip(*merged.sample()))
        for name, col in zip(self.FIELDS, columns):
            setattr(self, name, np.stack(col))
        self.batch = iter(columns[len(self.FIELDS):])  # (whatever a caller pushed beyond the six fields, as the reference's iterator leaves it)


class LoggerRL:
    """Episode / reward statistics of one sampling pass (logger_rl.py:4-68), fed from device tensors."""

    def __init__(self):
        self.num_steps = 0
        self.num_episodes = 0
        self.total_c_reward = 0.0
        self.total_c_info = 0.0  # one entry per reward term (5 or 6, by reward id)
        self.episode_c_rewards = []
        self.episode_lens = []
        self.sample_time = 0.0

    def add_steps(self, n, c_reward_sum, c_info_sum):
        self.num_steps += int(n)
        self.total_c_reward += float(c_reward_sum)
        self.total_c_info = self.total_c_info + np.asarray(c_info_sum, dtype=np.float64)

    def add_episodes(self, lens, c_rewards):
        self.num_episodes += len(lens)
        self.episode_lens += [int(x) for x in lens]
        self.episode_c_rewards += [float(x) for x in c_rewards]

    def end_sampling(self):
        n = max(self.num_steps, 1)
        self.avg_c_reward = self.total_c_reward / n
        self.avg_c_info = np.atleast_1d(self.total_c_info / n)
        self.avg_episode_len = float(np.mean(self.episode_lens)) if self.episode_lens else float(self.num_steps)
        self.avg_episode_c_reward = float(np.mean(self.episode_c_rewards)) if self.episode_c_rewards else self.total_c_reward
        self.max_c_reward = max(self.episode_c_rewards) if self.episode_c_rewards else self.total_c_reward
        self.min_c_reward = min(self.episode_c_rewards) if self.episode_c_rewards else self.total_c_reward

    @classmethod
    def merge(cls, loggers):
        out = cls()
        for lg in loggers:
            out.add_steps(lg.num_steps, lg.total_c_reward, lg.total_c_info)
            out.add_episodes(lg.episode_lens, lg.episode_c_rewards)
        out.end_sampling()
        return out
