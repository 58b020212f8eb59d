"""Self-collision rollout probe under a given kernel path (UHC_KERNEL_PATH = 0 chain | 1 general first | 2 sticky tiers): prints the bench sub-line."""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
args = bench.parse()
torch.cuda.set_device(0)
torch.set_default_dtype(torch.float64)
out = bench.rollout_probe(args, 0, torch.float64, "self_collision", warmup=8, steps=16, robot_cfg={"mesh": True, "model": "smpl"})
print(json.dumps({k: v for k, v in out.items() if k != "workload"}))
