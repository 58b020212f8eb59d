#!/bin/bash
# Round-5 GPU pass (through gpurun): tools/r05_pass.sh TAG STAGE [STAGE ...]; everything lands under gpurun_out/ with the tag, the summaries worth
# keeping are copied into profiles/ by hand.  Stages of its own:
#   parity200   tests/test_gpu_parity_200.py alone (north_star's 200-step bar on the bench's model classes); its per-run records land in
#               gpurun_out/parity200.jsonl and are copied to ${TAG}_parity200.jsonl
#   occupancy   tools/occupancy_sweep.sh (instrumented library): the fast tier's kernel at 1 / 2 / 3 / 4 workgroups per CU
#   tests_fast  the GPU suite without the 200-step file (which `parity200` runs)
#   soak        long synchronised rollouts of configs4 / ball_rollout / shapes (+ configs4 at 4096 envs): tier-4 env-steps, cap hits, dropped rows, step times
#   probes4096  the two ball-joint probes at 4096 envs (efc_overflow must stay 0)
#   ppo_prof    kernel trace of one PPO update (bench.py --no-probes --no-cpu-baseline --no-pgs-probe with a short rollout): which kernels the update spends its time in
# every other stage name is handed to tools/r04_pass.sh (tests, loop, bench, prof_headline, prof_floor, prof_configs4, prof_shapes, slowest, stage, meta).
set -u
TAG=${1:-r05_x}; shift
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/${TAG}
for stage in "$@"; do
case $stage in
parity200)
  rm -f gpurun_out/parity200.jsonl
  (timeout 1500 python -m pytest tests/test_gpu_parity_200.py -m gpu -q --tb=line -rs 2>&1 | grep -v amdgpu | tail -60) > ${O}_parity200_pytest.txt 2>&1
  cp gpurun_out/parity200.jsonl ${O}_parity200.jsonl 2>/dev/null
  tail -3 ${O}_parity200_pytest.txt ;;
occupancy)
  bash tools/occupancy_sweep.sh > ${O}_occupancy_sweep.txt 2>&1
  grep "===\|mean cycles" ${O}_occupancy_sweep.txt | cut -c1-200 ;;
parity_ball)
  rm -f gpurun_out/parity200.jsonl
  (timeout 1500 python -m pytest tests/test_gpu_parity_200.py -m gpu -q --tb=line -rs -k "ball" 2>&1 | grep -v amdgpu | tail -40) > ${O}_parity200_ball_pytest.txt 2>&1
  cp gpurun_out/parity200.jsonl ${O}_parity200_ball.jsonl 2>/dev/null
  tail -3 ${O}_parity200_ball_pytest.txt ;;
tier4)
  (timeout 900 python -m pytest tests/test_gpu_selfcollision.py -m gpu -q --tb=short -rs -s -k "tier_4 or solved_exactly or drops_rows or lying" 2>&1 | grep -v amdgpu | tail -60) > ${O}_tier4_pytest.txt 2>&1
  tail -25 ${O}_tier4_pytest.txt ;;
bench_tier4)
  python tools/bench_tier4.py 128 6 2>&1 | grep -v amdgpu > ${O}_bench_tier4.txt
  cat ${O}_bench_tier4.txt ;;
bench_tier4_prof)
  UHC_LIB=uhc_amd/csrc/libuhc_amd_prof.so python tools/bench_tier4.py 128 6 2>&1 | grep -v amdgpu > ${O}_bench_tier4_prof.txt
  cat ${O}_bench_tier4_prof.txt ;;
probe_configs4)
  python bench.py --only-probe configs4 > ${O}_probe_configs4.json 2> ${O}_probe_configs4.err
  python bench.py --only-probe ball_rollout > ${O}_probe_ball_rollout.json 2>> ${O}_probe_configs4.err
  cut -c1-1500 ${O}_probe_configs4.json ;;
soak)
  # stability of the tier chain with tier 4 and its consumers: long rollouts, every step synchronised (tools/diag_tier4_rollout.py SUMMARY=1)
  : > ${O}_soak.txt
  for w in configs4 ball_rollout; do  # (what tier 4 holds on these models: one line of the library's own log)
    (UHC_DEBUG=64 timeout 200 python tools/diag_tier4_rollout.py $w 64 2 2>&1 | grep "uhc tier 4" | sort -u | sed "s/^/$w: /") >> ${O}_soak.txt
  done
  for spec in "configs4 1024 1500" "ball_rollout 1024 1000" "shapes 1024 400" "configs4 4096 120"; do
    set -- $spec
    if [ "${SOAK:-full}" = short ] && [ "$1 $2" != "configs4 1024" ] && [ "$1" != ball_rollout ]; then continue; fi
    (SUMMARY=1 timeout 600 python tools/diag_tier4_rollout.py $1 $2 $3 2>&1 | grep -v amdgpu | tail -2; echo "rc=$?") >> ${O}_soak.txt
  done
  cut -c1-700 ${O}_soak.txt ;;
probes4096)
  # no dropped row at 4096 envs either (one repetition of 30 steps after 30: the queues of a general-tier-heavy workload are not tuned at that size)
  for pr in configs4 ball_rollout; do
    timeout 600 python bench.py --only-probe $pr --envs 4096 --probe-warmup 30 --probe-steps 30 --probe-reps 1 > ${O}_${pr}_4096.json 2>> ${O}_probe4096.err
    python - <<P
import json
d=json.load(open("${O}_${pr}_4096.json"))
print("${pr} @4096:", round(d["env_steps_per_s"]), "env-steps/s, ms", round(d["ms_per_step"],1), "overflow", d["efc_overflow_env_steps_all_reps"], "sweeps", d["sweeps_fallback_share_of_env_steps"], "tier4 share", round(d["tier4_primal_newton_share_of_env_steps"],5), "cap", d["tier4_newton_hit_its_cap_env_steps"], "nefc max", d["nefc_max_at_rep_ends"])
P
  done ;;
tests_fast)
  (timeout 1700 python -m pytest tests -m gpu -q --tb=short -rs --deselect tests/test_gpu_parity_200.py 2>&1 | grep -v amdgpu | tail -45) > ${O}_pytest_gpu.txt 2>&1
  tail -4 ${O}_pytest_gpu.txt ;;
ppo_prof)
  B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --preroll 6 --no-cpu-baseline --no-pgs-probe --no-probes"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo -o kt -- $B > $GRAFT_REPO_ROOT/${O}_bench_ppo_under_rocprof.json 2> /tmp/ppo.err)
  python tools/rocpd_summary.py /tmp/prof_ppo/kt_results.db ${O}_ppo_kernel_stats.txt "$TAG: rocprofv3 --kernel-trace --stats -- bench.py --steps 8 --warmup 2 --preroll 6 --no-cpu-baseline --no-pgs-probe --no-probes (16 rollout steps of 1024 envs + two full PPO updates over 16384 samples: the update's kernels are the Cijk_* / elementwise rows)" > /dev/null
  head -30 ${O}_ppo_kernel_stats.txt | cut -c1-180 ;;
*)
  bash tools/r04_pass.sh "$TAG" "$stage" ;;
esac
done
