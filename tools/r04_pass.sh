#!/bin/bash
# Round-4 GPU pass (through gpurun): tools/r04_pass.sh TAG STAGE [STAGE ...]; everything lands under gpurun_out/ with the tag, the
# summaries worth keeping are copied into profiles/ by hand.  Stages:
#   tests     pytest -m gpu (the whole suite)
#   loop      the ball-joint agent test (with its evaluation pass and fail-safe teleport) 20 x, each time as the first test of a fresh process
#   bench     python bench.py (driver-style: defaults)
#   prof_headline   kernel trace + PMC passes of the headline workload (self-colliding model class), tag _selfcol
#   prof_floor      kernel trace + VALU / FETCH / WRITE passes of --floor-only
#   prof_configs4   kernel trace + VALU / FETCH / WRITE passes of the configs[4] rollout probe, and of ball_rollout
#   prof_shapes     the same for the shapes probe (configs[3])
#   slowest         tools/diag_slowest.py on the headline and on configs4 (instrumented library): what the slowest general-tier env of a step does
#   stage     instrumented stage profiles (selfcol fast / general, ball_objects general)
#   meta      code-object metadata of every kernel
set -u
TAG=${1:-r04_x}; shift
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/${TAG}
pmc() {  # pmc NAME "COUNTERS" -- command...   -> /tmp/prof_NAME/p_results.db
  local name=$1 ctr=$2; shift 2; shift
  (cd /tmp && rocprofv3 --kernel-trace --pmc $ctr -d /tmp/prof_$name -o p -- "$@" > /tmp/$name.out 2> /tmp/$name.err) || tail -3 /tmp/$name.err
}
VALU="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
WAVE="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"
LDS="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
for stage in "$@"; do
case $stage in
tests)
  (timeout 1700 python -m pytest tests -m gpu -q --tb=short -rs 2>&1 | grep -v amdgpu | tail -45) > ${O}_pytest_gpu.txt 2>&1
  tail -4 ${O}_pytest_gpu.txt ;;
loop)
  : > ${O}_ball_agent_loop.txt
  ok=0; bad=0
  for i in $(seq 1 20); do
    if timeout 300 python -m pytest "tests/test_gpu_agent.py::test_agent_iteration_on_the_ball_joint_humanoid" -q -x > /tmp/loop_$i.txt 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); tail -30 /tmp/loop_$i.txt >> ${O}_ball_agent_loop.txt; fi
    echo "run $i: $(tail -1 /tmp/loop_$i.txt)" >> ${O}_ball_agent_loop.txt
  done
  echo "ball-joint agent iteration + evaluation (fail-safe teleport), first test of a fresh process: $ok passed, $bad failed of 20" | tee -a ${O}_ball_agent_loop.txt ;;
bench)
  python bench.py > ${O}_bench.json 2> ${O}_bench.err
  cut -c1-600 ${O}_bench.json ;;
prof_headline)
  B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ppo --no-pgs-probe --no-probes"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $B > $GRAFT_REPO_ROOT/${O}_bench_under_rocprof.json 2> /tmp/kt.err)
  python tools/rocpd_summary.py /tmp/prof_kt/kt_results.db ${O}_kernel_stats_selfcol.txt "$TAG: rocprofv3 --kernel-trace --stats -- bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ppo --no-pgs-probe --no-probes (headline: self-colliding model class)" > /dev/null
  P="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-ppo --no-pgs-probe --no-probes"
  pmc hf "FETCH_SIZE" -- $P;  python tools/pmc_summary.py /tmp/prof_hf/p_results.db ${O}_pmc_FETCH_SIZE_selfcol.txt "$TAG: --pmc FETCH_SIZE, headline (self-colliding model class), 51 control steps of 1024 envs (40 pre-roll + 3 + 8)" > /dev/null
  pmc hw "WRITE_SIZE" -- $P;  python tools/pmc_summary.py /tmp/prof_hw/p_results.db ${O}_pmc_WRITE_SIZE_selfcol.txt "$TAG: --pmc WRITE_SIZE, headline" > /dev/null
  pmc hv "$VALU" -- $P;       python tools/pmc_summary.py /tmp/prof_hv/p_results.db ${O}_pmc_VALU_F64_selfcol.txt "$TAG: --pmc $VALU, headline" > /dev/null
  pmc hc "$WAVE" -- $P;       python tools/pmc_summary.py /tmp/prof_hc/p_results.db ${O}_pmc_SQ_WAVE_CYCLES_selfcol.txt "$TAG: --pmc $WAVE, headline" > /dev/null
  pmc hl "$LDS" -- $P;        python tools/pmc_summary.py /tmp/prof_hl/p_results.db ${O}_pmc_SQ_INSTS_LDS_selfcol.txt "$TAG: --pmc $LDS, headline" > /dev/null
  python tools/pmc_alu.py ${O}_alu_headline.json 52224 /tmp/prof_hv/p_results.db /tmp/prof_hf/p_results.db /tmp/prof_hw/p_results.db -- "$TAG: headline (self-colliding model class), bench.py --steps 8 --warmup 3 behind the default 40-step pre-roll: 51 control steps x 1024 envs"
  grep "uhc_step" ${O}_kernel_stats_selfcol.txt | cut -c1-160 ;;
prof_floor)
  B="python $GRAFT_REPO_ROOT/bench.py --floor-only --steps 20 --warmup 5 --no-cpu-baseline --no-ppo --no-pgs-probe --no-probes"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_ktf -o kt -- $B > $GRAFT_REPO_ROOT/${O}_bench_floor_under_rocprof.json 2> /tmp/ktf.err)
  python tools/rocpd_summary.py /tmp/prof_ktf/kt_results.db ${O}_kernel_stats.txt "$TAG: rocprofv3 --kernel-trace --stats -- bench.py --floor-only --steps 20 --warmup 5 ... (static floor-only asset)" > /dev/null
  P="python $GRAFT_REPO_ROOT/bench.py --floor-only --steps 8 --warmup 3 --no-cpu-baseline --no-ppo --no-pgs-probe --no-probes"
  pmc ff "FETCH_SIZE" -- $P;  python tools/pmc_summary.py /tmp/prof_ff/p_results.db ${O}_pmc_FETCH_SIZE.txt "$TAG: --pmc FETCH_SIZE, --floor-only" > /dev/null
  pmc fw "WRITE_SIZE" -- $P;  python tools/pmc_summary.py /tmp/prof_fw/p_results.db ${O}_pmc_WRITE_SIZE.txt "$TAG: --pmc WRITE_SIZE, --floor-only" > /dev/null
  pmc fv "$VALU" -- $P;       python tools/pmc_summary.py /tmp/prof_fv/p_results.db ${O}_pmc_VALU_F64.txt "$TAG: --pmc $VALU, --floor-only" > /dev/null
  pmc fc "$WAVE" -- $P;       python tools/pmc_summary.py /tmp/prof_fc/p_results.db ${O}_pmc_SQ_WAVE_CYCLES.txt "$TAG: --pmc $WAVE, --floor-only" > /dev/null
  pmc fl "$LDS" -- $P;        python tools/pmc_summary.py /tmp/prof_fl/p_results.db ${O}_pmc_SQ_INSTS_LDS.txt "$TAG: --pmc $LDS, --floor-only" > /dev/null
  python tools/pmc_alu.py ${O}_alu_floor_only.json 52224 /tmp/prof_fv/p_results.db /tmp/prof_ff/p_results.db /tmp/prof_fw/p_results.db -- "$TAG: --floor-only, 51 control steps x 1024 envs (40-step pre-roll + 3 + 8)"
  grep "uhc_step" ${O}_kernel_stats.txt | cut -c1-160 ;;
prof_configs4|prof_shapes)
  if [ $stage = prof_shapes ]; then WL="shapes"; else WL="configs4 ball_rollout"; fi
  for w in $WL; do
    P="python $GRAFT_REPO_ROOT/bench.py --only-probe $w --probe-warmup 8 --probe-steps 12 --probe-reps 1"
    (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_$w -o kt -- $P > $GRAFT_REPO_ROOT/${O}_${w}_under_rocprof.json 2> /tmp/kt_$w.err)
    python tools/rocpd_summary.py /tmp/prof_kt_$w/kt_results.db ${O}_kernel_stats_$w.txt "$TAG: rocprofv3 --kernel-trace --stats -- bench.py --only-probe $w --probe-warmup 8 --probe-steps 12 --probe-reps 1" > /dev/null
    pmc ${w}f "FETCH_SIZE" -- $P; pmc ${w}w "WRITE_SIZE" -- $P; pmc ${w}v "$VALU" -- $P; pmc ${w}c "$WAVE" -- $P
    python tools/pmc_summary.py /tmp/prof_${w}v/p_results.db ${O}_pmc_VALU_F64_$w.txt "$TAG: --pmc $VALU, --only-probe $w" > /dev/null
    python tools/pmc_summary.py /tmp/prof_${w}c/p_results.db ${O}_pmc_SQ_WAVE_CYCLES_$w.txt "$TAG: --pmc $WAVE, --only-probe $w" > /dev/null
    python tools/pmc_summary.py /tmp/prof_${w}f/p_results.db ${O}_pmc_FETCH_SIZE_$w.txt "$TAG: --pmc FETCH_SIZE, --only-probe $w" > /dev/null
    python tools/pmc_summary.py /tmp/prof_${w}w/p_results.db ${O}_pmc_WRITE_SIZE_$w.txt "$TAG: --pmc WRITE_SIZE, --only-probe $w" > /dev/null
    python tools/pmc_alu.py ${O}_alu_$w.json 20480 /tmp/prof_${w}v/p_results.db /tmp/prof_${w}f/p_results.db /tmp/prof_${w}w/p_results.db -- "$TAG: --only-probe $w, 20 control steps x 1024 envs"
    grep "uhc_step" ${O}_kernel_stats_$w.txt | cut -c1-160
  done ;;
stage)
  MODEL=selfcol SOLVER=1 CAP=300 python tools/stage_profile.py 1024 10 > ${O}_stage_profile_selfcol.txt 2>&1
  SOLVER=1 python tools/stage_profile.py 1024 10 > ${O}_stage_profile.txt 2>&1
  MODEL=ball_objects SOLVER=1 CAP=300 UHC_FORCE_GENERAL=1 python tools/stage_profile.py 512 12 > ${O}_stage_profile_ball_objects_general.txt 2>&1 ;;
slowest)
  python tools/diag_slowest.py headline 40 50 2>&1 | grep -v amdgpu > ${O}_diag_slowest_headline.txt
  python tools/diag_slowest.py configs4 40 50 2>&1 | grep -v amdgpu > ${O}_diag_slowest_configs4.txt
  head -2 ${O}_diag_slowest_configs4.txt | cut -c1-400 ;;
meta)
  bash tools/kernel_meta.sh > ${O}_kernel_meta.txt 2>&1 ;;
*) echo "unknown stage $stage" ;;
esac
done
