cd $GRAFT_REPO_ROOT
echo "selfcol"; python tools/probe_selfcol.py 2>/dev/null | cut -c1-200
echo "ball_objects"; python bench.py --workload ball_objects --steps 60 --warmup 20 2>/dev/null | cut -c1-200
python tools/tier_trace.py gpurun_out/tier_trace_bo2.txt --workload ball_objects 2>&1 | grep -v amdgpu | grep -v "  env" | tail -24
timeout 1200 python -m pytest tests/test_gpu_selfcollision.py tests/test_gpu_env.py tests/test_gpu_behaviour.py tests/test_gpu_ball.py tests/test_gpu_agent.py -m gpu -q 2>&1 | grep -v amdgpu | tail -5
