"""Host side of the rollout bookkeeping kernels (uhc_amd/csrc/uhc_rollout.hip; C-ABI in include/uhc_amd.h): thin ctypes calls on torch
tensors.  torch only owns the memory and names the stream -- under HIP-graph capture that is the capturing stream, so the launches are
recorded like the framework's own.  Device float64 only; callers keep their torch path for CPU tensors and other dtypes."""
import ctypes as C

import torch

from ._lib import check, lib


def usable(*tensors) -> bool:
    return all(t is not None and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() for t in tensors)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def act(t_dev, state, mean, log_std, noise, mean_flags, states, actions, action):
    """states[:, t] = state; action = mean | mean + exp(log_std) * noise by mean_flags[t]; actions[:, t] = action."""
    n_env, T, obs_dim = states.shape
    act_dim = actions.shape[2]
    assert t_dev.dtype == torch.int64 and mean_flags.shape == (T, n_env) and mean.shape == (n_env, act_dim) and noise.shape == (n_env, act_dim)
    assert log_std.numel() == act_dim and action.shape == (n_env, act_dim) and state.shape == (n_env, obs_dim)
    check(lib().uhc_rollout_act(_stream(state), n_env, T, _p(t_dev), obs_dim, act_dim, _p(state), _p(mean), _p(log_std), _p(noise), _p(mean_flags),
                                _p(states), _p(actions), _p(action)))


def record(t_dev, reward, done, end, end_reward, parts, n_parts, rewards, dones, c_reward_sum, c_info_sum, redo=None, redo_counts=None):
    """rewards[:, t] = reward + end * end_reward; dones[:, t] = done; the pass's reward sums.  parts: [n_env][stride] (first n_parts count).
    redo / redo_counts: UHC_F_REDO of the step and the int64 [7] counters it is added to (general-tier env-steps, sweeps fallbacks, env-steps that lost rows beyond the last tier's capacity, large-tier / tier-4 env-steps, env-steps with a windowed exact solve, env-steps solved by Newton on the primal in tier 4, env-steps in which that iteration stopped at its cap)."""
    n_env, T = rewards.shape
    assert done.dtype == torch.int32 and end.dtype == torch.int32 and parts.dim() == 2 and parts.stride(1) == 1
    assert redo is None or (redo.dtype == torch.int32 and redo_counts is not None and redo_counts.dtype == torch.int64 and redo_counts.numel() == 7)
    check(lib().uhc_rollout_record(_stream(reward), n_env, T, _p(t_dev), _p(reward), _p(done), _p(end), _p(end_reward), _p(parts), parts.stride(0), n_parts,
                                   _p(rewards), _p(dones), _p(c_reward_sum), _p(c_info_sum), _p(redo), _p(redo_counts)))


def filter_scratch(n_rows, dim, device):
    return torch.empty(int(lib().uhc_filter_scratch_doubles(n_rows, dim)), dtype=torch.float64, device=device)


def filter_push(x, weights, n, mean, S, scratch):
    n_rows, dim = x.shape
    assert weights is None or (weights.dtype == torch.int32 and weights.is_contiguous() and weights.numel() == n_rows)
    check(lib().uhc_filter_push(_stream(x), _p(x), n_rows, dim, _p(weights), _p(n), _p(mean), _p(S), _p(scratch)))


def filter_apply(x, n, mean, S, demean, destd, clip, out, t_inc=None):
    n_rows, dim = x.shape
    assert out.shape == x.shape and (t_inc is None or t_inc.dtype == torch.int64)
    check(lib().uhc_filter_apply(_stream(x), _p(x), n_rows, dim, _p(n), _p(mean), _p(S), int(bool(demean)), int(bool(destd)), float(clip or 0.0), _p(out), _p(t_inc)))
