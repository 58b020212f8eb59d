cd $GRAFT_REPO_ROOT
# A/B of the sticky-tier scheduling switches on the self-colliding rollout and the ball_objects scene
# (UHC_DEBUG bit 2: no resume at the hand-on substep, bit 3: fast tier in env order, bit 5: no gate before the fast tier's launch)
for d in 0 8 4; do echo "UHC_DEBUG=$d"; UHC_DEBUG=$d python tools/probe_selfcol.py 2>/dev/null | cut -c1-260; done
python tools/tier_trace.py gpurun_out/tier_trace_f.txt 2>&1 | grep -v amdgpu | tail -28
for d in 0; do echo "UHC_DEBUG=$d"; UHC_DEBUG=$d python bench.py --workload ball_objects --steps 60 --warmup 20 2>/dev/null | cut -c1-200; done
