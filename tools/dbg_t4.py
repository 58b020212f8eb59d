import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from uhc_amd import sim as S
sys.argv = ["bench.py"]
args = bench.parse()
torch.cuda.set_device(0); torch.set_default_dtype(torch.float64)
agent = bench.build_agent(args, 0, 0, torch.float64, robot_cfg={"mesh": True, "model": "smpl"})
agent.per_epoch_update(0)
env = agent.env
agent.rollout_begin(80)
for t in range(60):
    agent.rollout_step()
    if t >= 50:
        torch.cuda.synchronize()
        tier = env.sim.field(S.F_TIER).cpu().numpy(); redo = env.sim.field(S.F_REDO).cpu().numpy(); nefc = env.sim.field(S.F_NEFC).cpu().numpy()
        print("step", t, "next tiers", np.bincount(tier, minlength=5).tolist(), "primal", int(((redo >> 30) & 1).sum()), "general/large", int((redo & 1).sum()), "nefc max", int(nefc.max()), "nefc of tier>=2", sorted(nefc[(redo & 1) != 0].tolist())[-8:], flush=True)
