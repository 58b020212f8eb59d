#!/bin/bash
# A/B of tier 4's own queue consumers on one box: tools/r05_q4_ab.sh TAG  (tier-4 tests first, under a watchdog; then the two ball-joint probes with and
# without the consumers -- UHC_Q4_MAX=0 leaves what the large tier hands on for the chained launch at the end of the step)
set -u
TAG=${1:-r05_q4}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/${TAG}
(UHC_DEBUG=64 timeout 240 python -m pytest tests/test_gpu_selfcollision.py -m gpu -q -x --tb=short -s -k "sticky_queues_hand" 2>&1 | grep -v amdgpu | grep "tier-4 consumers\|passed\|failed\|sticky queues\|rror" | head -12) > ${O}_sticky_test.txt 2>&1
cat ${O}_sticky_test.txt
grep -q "1 passed" ${O}_sticky_test.txt || { echo "sticky tier-4 test did not pass: stopping"; exit 1; }
(timeout 600 python -m pytest tests/test_gpu_selfcollision.py -m gpu -q --tb=short -s -k "tier_4 or solved_exactly or drops_rows" 2>&1 | grep -v amdgpu | tail -25) > ${O}_tier4_pytest.txt 2>&1
tail -8 ${O}_tier4_pytest.txt
for q in 16 0 16; do
  UHC_Q4_MAX=$q timeout 300 python bench.py --only-probe configs4 > ${O}_configs4_q${q}.json 2>> ${O}_probe.err
  python - <<P
import json
d=json.load(open("${O}_configs4_q${q}.json"))
print("configs4 UHC_Q4_MAX=${q}:", round(d["env_steps_per_s"]), d["env_steps_per_s_each_rep"], "ms", round(d["ms_per_step"],2), "tier4 share", d["tier4_primal_newton_share_of_env_steps"], "overflow", d["efc_overflow_env_steps_all_reps"], "sweeps", d["sweeps_fallback_share_of_env_steps"], "cap", d["tier4_newton_hit_its_cap_env_steps"])
P
done
UHC_Q4_MAX=16 timeout 300 python bench.py --only-probe ball_rollout > ${O}_ball_q16.json 2>> ${O}_probe.err
python - <<P
import json
d=json.load(open("${O}_ball_q16.json"))
print("ball_rollout UHC_Q4_MAX=16:", round(d["env_steps_per_s"]), d["env_steps_per_s_each_rep"], "ms", round(d["ms_per_step"],2), "tier4 share", d["tier4_primal_newton_share_of_env_steps"], "overflow", d["efc_overflow_env_steps_all_reps"])
P
tail -5 ${O}_probe.err
