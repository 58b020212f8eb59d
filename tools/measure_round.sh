#!/bin/bash
# Run on the GPU box (through gpurun): the round's measurement pass.  Everything lands under gpurun_out/ with the given tag; the
# summaries worth keeping are then copied into profiles/ by hand.
set -u
TAG=${1:-r02_v4}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v amdgpu | tail -15) > gpurun_out/${TAG}_pytest_gpu.txt 2>&1
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --workload ball_objects --steps 40 --warmup 20 > gpurun_out/${TAG}_ball_objects.json 2> /dev/null
python bench.py --workload ball_objects --fixed-path --steps 40 --warmup 20 > gpurun_out/${TAG}_ball_objects_fixed_path.json 2> /dev/null
python bench.py --workload ball_objects --general-only --steps 40 --warmup 20 > gpurun_out/${TAG}_ball_objects_general_only.json 2> /dev/null
python bench.py --workload ball_objects --objects 0 --steps 40 --warmup 20 > gpurun_out/${TAG}_ball_objects_no_objects.json 2> /dev/null
python bench.py --shapes 1023 --no-cpu-baseline --no-ppo > gpurun_out/${TAG}_bench_shapes.json 2> gpurun_out/${TAG}_bench_shapes.err
SOLVER=1 python tools/stage_profile.py 1024 10 > gpurun_out/${TAG}_stage_profile.txt 2>&1
MODEL=ball_objects SOLVER=1 CAP=300 UHC_FORCE_GENERAL=1 python tools/stage_profile.py 256 12 > gpurun_out/${TAG}_stage_profile_ball_objects_general.txt 2>&1
bash tools/profile_on_gpu.sh $TAG > gpurun_out/${TAG}_profile_log.txt 2>&1
tail -3 gpurun_out/${TAG}_pytest_gpu.txt
cut -c1-400 gpurun_out/${TAG}_bench.json
