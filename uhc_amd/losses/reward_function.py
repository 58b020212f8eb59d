"""Imitation rewards, by the reference's ids (uhc/losses/reward_function.py:823-833).

On this build the reward is evaluated inside the env step kernel (uhc_amd/csrc/uhc_env.hip, which restates
world_rfc_implicit_reward (reward_function.py:12-88; world_rfc_implicit_reward_quat, :92-171, is the same body),
world_rfc_explicit_reward (:253-341), world_rfc_implicit_v1_mul (:174-250), world_rfc_explicit_mul_reward (:346-430),
world_rfc_implicit_v2 (:643-723) and world_rfc_implicit_v3 (:726-820)) so that rollout buffers never leave HBM; the
functions here keep the reference's call signature ``f(env, state, action, info) -> (reward, components)`` and read
the value the kernel produced for the facade env's last step.  The local_rfc_* ids (:433-640) are not built."""
import numpy as np

from .._capi import REWARD_IDS


def _device_reward(env, state, action, info):
    r, parts = env.last_reward
    return r, np.asarray(parts)


def _named(name):
    def f(env, state, action, info):
        return _device_reward(env, state, action, info)
    f.__name__ = name
    return f


world_rfc_implicit_reward = _named("world_rfc_implicit_reward")
world_rfc_implicit_reward_quat = _named("world_rfc_implicit_reward_quat")
world_rfc_explicit_reward = _named("world_rfc_explicit_reward")
world_rfc_implicit_v1_mul = _named("world_rfc_implicit_v1_mul")
world_rfc_explicit_mul_reward = _named("world_rfc_explicit_mul_reward")
world_rfc_implicit_v2 = _named("world_rfc_implicit_v2")
world_rfc_implicit_v3 = _named("world_rfc_implicit_v3")

reward_func = {
    "world_rfc_implicit": world_rfc_implicit_reward,
    "world_rfc_implicit_quat": world_rfc_implicit_reward_quat,
    "world_rfc_implicit_v1_mul": world_rfc_implicit_v1_mul,
    "world_rfc_explicit": world_rfc_explicit_reward,
    "world_rfc_explicit_mul": world_rfc_explicit_mul_reward,
    "world_rfc_implicit_v2": world_rfc_implicit_v2,
    "world_rfc_implicit_v3": world_rfc_implicit_v3,
}
DEVICE_REWARD_IDS = tuple(REWARD_IDS)
