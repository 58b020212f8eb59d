#!/bin/bash
# Run on the GPU box (through gpurun): the round's measurement pass.  Everything lands under gpurun_out/ with the given tag; the
# summaries worth keeping are then copied into profiles/ by hand.  tools/measure_round.sh TAG [notests]
set -u
TAG=${1:-r03_v1}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
if [ "${2:-}" != "notests" ]; then
  (timeout 1500 python -m pytest tests -m gpu -q --tb=short -rs 2>&1 | grep -v amdgpu | tail -25) > gpurun_out/${TAG}_pytest_gpu.txt 2>&1
  (timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v amdgpu | tail -60) > gpurun_out/${TAG}_pytest_gpu_second_run.txt 2>&1  # (the queues are concurrent code: the suite runs twice)
fi
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --workload ball_objects --steps 120 --warmup 20 > gpurun_out/${TAG}_ball_objects.json 2> /dev/null
python bench.py --workload ball_objects --general-only --steps 40 --warmup 20 > gpurun_out/${TAG}_ball_objects_general_first.json 2> /dev/null
python bench.py --workload ball_objects --objects 0 --steps 40 --warmup 20 > gpurun_out/${TAG}_ball_objects_no_objects.json 2> /dev/null
python bench.py --shapes 1023 --no-cpu-baseline --no-ppo --no-pgs-probe > gpurun_out/${TAG}_bench_shapes.json 2> gpurun_out/${TAG}_bench_shapes.err
python bench.py --envs 4096 --no-cpu-baseline --no-pgs-probe --no-probes > gpurun_out/${TAG}_bench_envs4096.json 2> /dev/null
SOLVER=1 python tools/stage_profile.py 1024 10 > gpurun_out/${TAG}_stage_profile.txt 2>&1
MODEL=selfcol SOLVER=1 CAP=300 python tools/stage_profile.py 1024 10 > gpurun_out/${TAG}_stage_profile_selfcol.txt 2>&1
MODEL=selfcol SOLVER=1 CAP=300 UHC_FORCE_GENERAL=1 python tools/stage_profile.py 512 10 > gpurun_out/${TAG}_stage_profile_selfcol_general.txt 2>&1
MODEL=ball_objects SOLVER=1 CAP=300 UHC_FORCE_GENERAL=1 python tools/stage_profile.py 512 12 > gpurun_out/${TAG}_stage_profile_ball_objects_general.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d /tmp/prof_sc -o kt -- python tools/probe_selfcol.py > gpurun_out/${TAG}_selfcol_probe_under_rocprof.json 2>/tmp/sc.err
python tools/timeline.py /tmp/prof_sc/kt_results.db gpurun_out/${TAG}_selfcol_timeline.txt 3 > /dev/null
python tools/rocpd_summary.py /tmp/prof_sc/kt_results.db gpurun_out/${TAG}_selfcol_kernel_stats.txt "$TAG: rocprofv3 --kernel-trace --stats -- python tools/probe_selfcol.py (self-collision rollout, sticky tiers)" > /dev/null
python tools/tier_trace.py gpurun_out/${TAG}_tier_trace_selfcol.txt > /dev/null 2>&1
python tools/tier_trace.py gpurun_out/${TAG}_tier_trace_ball_objects.txt --workload ball_objects > /dev/null 2>&1
bash tools/profile_on_gpu.sh $TAG > gpurun_out/${TAG}_profile_log.txt 2>&1
bash tools/kernel_meta.sh > gpurun_out/${TAG}_kernel_meta.txt 2>&1
tail -3 gpurun_out/${TAG}_pytest_gpu.txt
cut -c1-400 gpurun_out/${TAG}_bench.json
