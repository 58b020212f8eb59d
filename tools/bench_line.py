"""Reads bench.py JSON lines on stdin and prints the few numbers a layout / knob sweep compares (value, ms per step, fast-tier kernel ms,
env-steps through the general / large tier)."""
import json
import sys

for line in sys.stdin:
    try:
        d = json.loads(line)
    except Exception:
        continue
    if "value" in d:
        print(round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["kernel_ms"], 3), d["workload_stats"]["general_or_large_tier_env_steps_timed_region"])
    else:
        print(round(d["env_steps_per_s"]), round(d["ms_per_step"], 3), round(d["first_tier_kernel_ms"], 3), d["general_or_large_tier_share_of_env_steps"])
