cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-ppo --no-pgs-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k in ('self_collision','shapes','ball_rollout','ball_objects'): print(k, round(d[k]['env_steps_per_s']), d[k]['ms_per_step'])"
TRACE_BALL=1 UHC_DEBUG=64 python tools/tier_trace.py gpurun_out/tier_trace_ball2.txt 2>&1 | grep -v amdgpu | grep "^## \|gave up [1-9]"
