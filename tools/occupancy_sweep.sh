#!/bin/bash
# What bounds the fast tier's wave (VERDICT r4 "next" 4)?  The SAME kernel (self-colliding model class, exact solver) at 1 / 2 / 3 / 4 workgroups per
# CU: the workgroup's LDS request is padded (UHC_LDS_PAD_FAST; 4 per CU = the 40 KiB layout) and the batch is 256 x that many envs, so every
# workgroup is resident at once and every CU holds exactly `occ` of them.  Read: mean cycles per env-step flat in occ => the wave is latency-bound and
# a second wave per SIMD would be free throughput; rising with occ => the co-resident waves contend for a CU-shared resource (LDS bandwidth,
# scratch through L1 / TA, the scalar cache).  Instrumented build (cycle counters per stage), plain tier chain, 10 control steps from the standing pose.
#   tools/occupancy_sweep.sh > gpurun_out/r05_occupancy_sweep.txt
cd "$(dirname "$0")/.."
export MODEL=selfcol SOLVER=1 CAP=300
run() {  # occ envs [env assignments...]
  local occ=$1 envs=$2; shift 2
  echo "=== $occ workgroup(s) per CU, $envs envs  ($*)"
  env "$@" python tools/stage_profile.py $envs 10 2>&1 | grep -v amdgpu | head -24
}
run 1 256 UHC_LDS_PAD_FAST=100
run 2 512 UHC_LDS_PAD_FAST=64
run 3 768 UHC_LDS_PAD_FAST=52
run 4 1024 UHC_FAST_DENSE=40,6
echo "=== the same at a fixed 256 envs (one workgroup per CU whatever the padding): the spread between runs"
run 1 256 UHC_LDS_PAD_FAST=52
