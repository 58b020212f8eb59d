#!/bin/bash
# A/B of two builds of the library on one box: tools/r05_ab_lib.sh TAG  -- tier-4 tests on the new one, then configs4 / ball_rollout probes alternating
# between uhc_amd/csrc/libuhc_amd_prev.so (a copy of the previous build, travels with the snapshot) and the new libuhc_amd.so
set -u
TAG=${1:-r05_ab}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/${TAG}
(timeout 600 python -m pytest tests/test_gpu_selfcollision.py -m gpu -q --tb=short -s -k "tier_4 or solved_exactly or drops_rows" 2>&1 | grep -v amdgpu | tail -25) > ${O}_tier4_pytest.txt 2>&1
tail -4 ${O}_tier4_pytest.txt | cut -c1-400
grep -q " passed" ${O}_tier4_pytest.txt || exit 1
grep -q " failed" ${O}_tier4_pytest.txt && exit 1
python tools/bench_tier4.py 128 4 2>&1 | grep -v amdgpu > ${O}_bench_tier4_new.txt; head -3 ${O}_bench_tier4_new.txt | cut -c1-300
UHC_LIB=$PWD/uhc_amd/csrc/libuhc_amd_prev.so python tools/bench_tier4.py 128 4 2>&1 | grep -v amdgpu > ${O}_bench_tier4_prev.txt; head -3 ${O}_bench_tier4_prev.txt | cut -c1-300
for v in new prev new; do
  for pr in configs4 ball_rollout; do
    if [ $v = prev ]; then export UHC_LIB=$PWD/uhc_amd/csrc/libuhc_amd_prev.so; else unset UHC_LIB; fi
    timeout 300 python bench.py --only-probe $pr > ${O}_${pr}_${v}.json 2>> ${O}_probe.err
    python - <<P
import json
d=json.load(open("${O}_${pr}_${v}.json"))
print("${pr} ${v}:", round(d["env_steps_per_s"]), [round(x) for x in d["env_steps_per_s_each_rep"]], "ms", round(d["ms_per_step"],2), "tier4 share", round(d["tier4_primal_newton_share_of_env_steps"],5), "overflow", d["efc_overflow_env_steps_all_reps"], "sweeps", d["sweeps_fallback_share_of_env_steps"], "cap", d["tier4_newton_hit_its_cap_env_steps"])
P
  done
done
