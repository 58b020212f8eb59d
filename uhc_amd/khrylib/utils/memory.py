"""`Memory` -- the per-worker transition store of the reference's CPU sampler (uhc/khrylib/utils/memory.py:4-23: push / sample / append /
len), kept for callers written against it (the reference's `AgentCopycat.sample_worker` pushes one tuple per step and `TrajBatch` stacks the
fields).  This build's own sampler does not go through it: its rollout buffers are device tensors written by `uhc_rollout_act` /
`uhc_rollout_record` (uhc_amd/khrylib/rl/agents: `RolloutBatch`)."""
import random


class Memory:
    def __init__(self):
        self.memory = []

    def push(self, *fields):
        self.memory.append(list(fields))

    def sample(self, batch_size=None):
        return self.memory if batch_size is None else random.sample(self.memory, batch_size)

    def append(self, other):
        self.memory.extend(other.memory)

    def __len__(self):
        return len(self.memory)
