#!/bin/bash
# Round-6 GPU pass (through gpurun): tools/r06_pass.sh TAG STAGE [STAGE ...]; everything lands under gpurun_out/ with the tag, the summaries worth
# keeping are copied into profiles/ by hand.  Stages of its own (every other stage name is handed to tools/r05_pass.sh, which hands on to r04_pass.sh):
#   t4_tests     the tests that run tier 4 (one-workgroup-per-env form AND the four-wave queue consumers) against the oracle
#   t4_bench     tools/bench_tier4.py on the new library and on the round-5 one (libuhc_amd_r05.so) -- same box
#   t4_prof      the same scene on the instrumented libraries (stage cycles of the Newton iteration, new and round 5)
#   t4_diag      tools/diag_slowest.py SELECT=tier4 configs4 on the instrumented libraries: where a tier-4 env-step of the ROLLOUT goes
#   ab_probes    configs4 / ball_rollout probes alternating between the new library and the round-5 one (boxes differ by up to a quarter: only a same-box A/B counts)
#   ab_headline  the headline (bench.py --no-probes ...) alternating between the two libraries
set -u
TAG=${1:-r06_x}; shift
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/${TAG}
OLD=$PWD/uhc_amd/csrc/libuhc_amd_r05.so
OLDP=$PWD/uhc_amd/csrc/libuhc_amd_prof_r05.so
NEWP=$PWD/uhc_amd/csrc/libuhc_amd_prof.so
probe_line() {  # probe_line FILE LABEL
python - "$1" "$2" <<P
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2] + ":", round(d["env_steps_per_s"]), [round(x) for x in d["env_steps_per_s_each_rep"]], "ms", round(d["ms_per_step"], 2), "tier4 share", round(d.get("tier4_primal_newton_share_of_env_steps", 0), 5),
      "overflow", d.get("efc_overflow_env_steps_all_reps"), "sweeps", d.get("sweeps_fallback_share_of_env_steps"), "cap", d.get("tier4_newton_hit_its_cap_env_steps"))
P
}
for stage in "$@"; do
case $stage in
t4_tests)
  (timeout 700 python -m pytest tests/test_gpu_selfcollision.py tests/test_gpu_env_objects.py -m gpu -q --tb=short -rs -s -k "tier_4 or solved_exactly or drops_rows or lying or objects" 2>&1 | grep -v amdgpu | tail -60) > ${O}_tier4_pytest.txt 2>&1
  tail -30 ${O}_tier4_pytest.txt | cut -c1-300 ;;
t4_bench)
  timeout 300 python tools/bench_tier4.py 128 6 2>&1 | grep -v amdgpu > ${O}_bench_tier4_new.txt; cat ${O}_bench_tier4_new.txt | cut -c1-260
  UHC_LIB=$OLD timeout 300 python tools/bench_tier4.py 128 6 2>&1 | grep -v amdgpu > ${O}_bench_tier4_r05.txt; cat ${O}_bench_tier4_r05.txt | cut -c1-260 ;;
t4_prof)
  cmp -s $NEWP $OLDP || UHC_LIB=$NEWP timeout 300 python tools/bench_tier4.py 128 6 2>&1 | grep -v amdgpu > ${O}_bench_tier4_prof_new.txt; tail -32 ${O}_bench_tier4_prof_new.txt
  [ -f $OLDP ] && { UHC_LIB=$OLDP timeout 300 python tools/bench_tier4.py 128 6 2>&1 | grep -v amdgpu > ${O}_bench_tier4_prof_r05.txt; tail -32 ${O}_bench_tier4_prof_r05.txt; } ;;
t4_diag)
  cmp -s $NEWP $OLDP || UHC_LIB=$NEWP SELECT=tier4 timeout 600 python tools/diag_slowest.py configs4 30 50 2>&1 | grep -v amdgpu > ${O}_diag_tier4_configs4_new.txt; tail -42 ${O}_diag_tier4_configs4_new.txt | cut -c1-200
  [ -f $OLDP ] && { UHC_LIB=$OLDP SELECT=tier4 timeout 600 python tools/diag_slowest.py configs4 30 50 2>&1 | grep -v amdgpu > ${O}_diag_tier4_configs4_r05.txt; tail -42 ${O}_diag_tier4_configs4_r05.txt | cut -c1-200; } ;;
ab_probes)
  for v in new r05 new r05; do
    for pr in ${PROBES:-configs4 ball_rollout}; do
      if [ $v = r05 ]; then export UHC_LIB=$OLD; else unset UHC_LIB; fi
      timeout 300 python bench.py --only-probe $pr > ${O}_${pr}_${v}.json 2>> ${O}_probe.err
      probe_line ${O}_${pr}_${v}.json "$pr $v"
    done
  done
  unset UHC_LIB ;;
ab_headline)
  for v in new r05 new r05; do
    if [ $v = r05 ]; then export UHC_LIB=$OLD; else unset UHC_LIB; fi
    timeout 300 python bench.py --no-probes --no-cpu-baseline --no-ppo --no-pgs-probe > ${O}_headline_${v}.json 2>> ${O}_probe.err
    python -c "import json,sys; d=json.load(open('${O}_headline_${v}.json')); print('headline $v:', round(d['value']), 'ms', round(d['ms_per_step'],2), 'kernel ms', d['roofline'].get('kernel_ms_per_launch'))"
  done
  unset UHC_LIB ;;
sweep_t4rows)
  # UHC_T4_ROWS (KernelArgs::t4_rows): from how many rows on an env of the general / large tier starts its NEXT step in tier 4 (0: never)
  for v in ${T4ROWS:-0 96 128 160 200 256}; do
    for pr in ${PROBES:-configs4 ball_rollout}; do
      UHC_T4_ROWS=$v timeout 300 python bench.py --only-probe $pr --probe-reps 2 > ${O}_t4rows${v}_${pr}.json 2>> ${O}_probe.err
      probe_line ${O}_t4rows${v}_${pr}.json "UHC_T4_ROWS=$v $pr"
    done
    UHC_T4_ROWS=$v timeout 300 python bench.py --steps 40 --warmup 5 --no-probes --no-cpu-baseline --no-ppo --no-pgs-probe > ${O}_t4rows${v}_headline.json 2>> ${O}_probe.err
    python -c "import json,sys; d=json.load(open('${O}_t4rows${v}_headline.json')); w=d['workload_stats']; print('UHC_T4_ROWS=$v headline:', round(d['value']), 'ms', round(d['ms_per_step'],2), 'kernel ms', round(d['roofline']['kernel_ms'],2), 'tier4 env-steps', w['tier4_primal_newton_env_steps_timed_region'], 'general/large', w['general_or_large_tier_env_steps_timed_region'])"
  done ;;
*)
  bash tools/r05_pass.sh "$TAG" "$stage" ;;
esac
done
