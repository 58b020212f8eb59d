#!/bin/bash
# A/B of the dense fast tier's LDS layout (UHC_FAST_DENSE = "KiB,body-body row slots,contacts"): the headline workload and, with PROBES set,
# rollout probes.   tools/ab_layout.sh TAG "52,12,16" "52,12,32" ...      PROBES="configs4 ball_rollout" tools/ab_layout.sh ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; TAG=$1; shift
O=gpurun_out/${TAG}_ab_layout.txt; : > $O
B="python bench.py --steps 60 --warmup 40 --no-probes --no-cpu-baseline --no-ppo --no-pgs-probe"
one() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
if "value" in d:
    w = d["workload_stats"]
    print(f"{sys.argv[2]:>12s} {sys.argv[1]:>10s}: {d['value']:9.0f} env-steps/s  {d['ms_per_step']:6.2f} ms/step  fast-tier kernel {d['roofline']['kernel_ms']:6.2f} ms  general/large tier env-steps {w['general_or_large_tier_env_steps_timed_region']} ({w['large_tier_env_steps_timed_region']} large)  overflow {w['efc_overflow_env_steps_timed_region']}  nefc max {w['nefc_max']}")
else:
    print(f"{sys.argv[2]:>12s} {sys.argv[1]:>10s}: {d['env_steps_per_s']:9.0f} env-steps/s  {d['ms_per_step']:6.2f} ms/step  fast-tier kernel {d['first_tier_kernel_ms']:6.2f} ms  general/large share {d['general_or_large_tier_share_of_env_steps']:.3f} (large {d['large_tier_share_of_env_steps']:.3f})  overflow env-steps {d['efc_overflow_env_steps']}  sweeps {d['sweeps_fallback_share_of_env_steps']:.5f}")
PY
}
for L in "$@"; do
  UHC_FAST_DENSE=$L $B > /tmp/ab.json 2>/dev/null; one headline "$L" >> $O
  for P in ${PROBES:-}; do UHC_FAST_DENSE=$L python bench.py --only-probe $P --probe-reps 1 --probe-steps 60 --probe-warmup 40 > /tmp/ab.json 2>/dev/null; one $P "$L" >> $O; done
done
cat $O
