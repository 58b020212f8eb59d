#!/bin/bash
# One measurement pass on the GPU box (through gpurun): tools/gpu_pass.sh TAG [notests].  Everything lands under gpurun_out/ with the tag.
set -u
TAG=${1:-r03_x}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
if [ "${2:-}" != "notests" ]; then
  (timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v amdgpu | tail -40) > gpurun_out/${TAG}_pytest_gpu.txt 2>&1
fi
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
MODEL=selfcol SOLVER=1 CAP=300 python tools/stage_profile.py 1024 10 > gpurun_out/${TAG}_stage_profile_selfcol.txt 2>&1
MODEL=selfcol SOLVER=1 CAP=300 UHC_FORCE_GENERAL=1 python tools/stage_profile.py 512 10 > gpurun_out/${TAG}_stage_profile_selfcol_general.txt 2>&1
MODEL=ball_objects SOLVER=1 CAP=300 UHC_FORCE_GENERAL=1 python tools/stage_profile.py 512 12 > gpurun_out/${TAG}_stage_profile_ball_objects_general.txt 2>&1
tail -4 gpurun_out/${TAG}_pytest_gpu.txt 2>/dev/null
cut -c1-300 gpurun_out/${TAG}_bench.json
