#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + separate PMC passes of the bench workload; only small
# text summaries are left under gpurun_out/ (the raw rocpd databases go to /tmp).
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
BENCH="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ppo --no-pgs-probe --no-probes"
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $BENCH > gpurun_out/${TAG}_bench_under_rocprof.json 2> /tmp/kt.err
python tools/rocpd_summary.py /tmp/prof_kt/kt_results.db gpurun_out/${TAG}_kernel_stats.txt "$TAG: rocprofv3 --kernel-trace --stats -- $BENCH" > /dev/null
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass -d /tmp/prof_$name -o p -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-ppo --no-pgs-probe --no-probes > /dev/null 2> /tmp/$name.err
  python tools/pmc_summary.py /tmp/prof_$name/p_results.db gpurun_out/${TAG}_pmc_$name.txt "$TAG: rocprofv3 --kernel-trace --pmc $pass" > /dev/null
done
# float64 VALU instruction mix of the fused step kernel (for the counter-based ALU roofline of bench.py)
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*" | sort -u | tr "\n" " " > gpurun_out/${TAG}_valu_counters_available.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d /tmp/prof_f64 -o p -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-ppo --no-pgs-probe --no-probes > /dev/null 2> /tmp/f64.err
python tools/pmc_summary.py /tmp/prof_f64/p_results.db gpurun_out/${TAG}_pmc_VALU_F64.txt "$TAG: rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" > /dev/null || tail -3 /tmp/f64.err
# the PPO update (GAE + 10 epochs) on the samples of a short rollout: which kernels the learner time goes to
rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo -o kt -- python bench.py --steps 24 --warmup 1 --no-cpu-baseline --no-pgs-probe --no-probes > gpurun_out/${TAG}_bench_ppo_under_rocprof.json 2> /tmp/ppo.err
python tools/rocpd_summary.py /tmp/prof_ppo/kt_results.db gpurun_out/${TAG}_ppo_kernel_stats.txt "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py --steps 24 --warmup 1 --no-cpu-baseline (rollout + one PPO update)" > /dev/null
tail -1 gpurun_out/${TAG}_bench_under_rocprof.json | cut -c1-300
cat gpurun_out/${TAG}_kernel_stats.txt | head -8 | cut -c1-200
cat gpurun_out/${TAG}_pmc_*.txt | grep -v "^#" | cut -c1-200
