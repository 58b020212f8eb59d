#!/bin/bash
# the debug library (poisoned LDS + guard words behind every LDS region) under the agent-level tests and N fresh-process runs of the ball-joint agent
# test: tools/r05_guard.sh TAG [N]   ->  gpurun_out/TAG_guard.txt
set -u
TAG=${1:-r05_g}; N=${2:-8}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/${TAG}_guard.txt
export UHC_LIB=$PWD/uhc_amd/csrc/libuhc_amd_poison.so UHC_GUARD_LDS=1
: > $O
timeout 900 python -m pytest tests/test_gpu_agent.py tests/test_gpu_env_objects.py tests/test_gpu_selfcollision.py tests/test_gpu_ball.py -m gpu -q -s --tb=short > /tmp/guard_suite.txt 2>&1
echo "agent-level + objects + self-collision + ball-joint tests on the debug library: $(tail -1 /tmp/guard_suite.txt)" >> $O
echo "  batches created with guard words: $(grep -c 'uhc guard: LDS guard words on' /tmp/guard_suite.txt); freed: $(grep -c 'uhc guard: batch of' /tmp/guard_suite.txt); lines reporting an overwritten guard word: $(grep -c OVERWRITTEN /tmp/guard_suite.txt)" >> $O
echo "  fenced HBM arrays: $(grep 'fenced HBM arrays checked' /tmp/guard_suite.txt | awk '{n+=$3; b+=$8} END {print n " checked over all batches, " b " with an overwritten fence"}')" >> $O
grep -m3 "uhc guard: LDS guard words on" /tmp/guard_suite.txt | sort -u >> $O
grep -m5 OVERWRITTEN /tmp/guard_suite.txt >> $O
grep -i "failed\|error" /tmp/guard_suite.txt | head -5 >> $O
ok=0; bad=0; hits=0
for i in $(seq 1 $N); do
  if timeout 300 python -m pytest "tests/test_gpu_agent.py::test_agent_iteration_on_the_ball_joint_humanoid" -q -x -s > /tmp/gloop_$i.txt 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); tail -20 /tmp/gloop_$i.txt >> $O; fi
  hits=$((hits + $(grep -c OVERWRITTEN /tmp/gloop_$i.txt)))
done
echo "ball-joint agent iteration + evaluation (fail-safe teleport) as the first test of a fresh process, debug library: $ok passed, $bad failed of $N; guard words overwritten: $hits" >> $O
cat $O
