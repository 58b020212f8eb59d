#!/bin/bash
# Run on the GPU box: A/B of the sticky-tier scheduling switches on the self-colliding rollout (tools/probe_selfcol.py = the bench line
# `self_collision`).  UHC_DEBUG bits: 2 (4) a handed-on env restarts its step instead of resuming at the substep, 3 (8) the fast tier launches
# in env order, 5 (32) no gate before the fast tier's launch, 7 (128) no box cull of the convex pairs.  UHC_FAST_DENSE=KiB,rows: layout
# of the dense fast tier.  Results of round 3: profiles/r03_ab_switches.txt.
cd "$GRAFT_REPO_ROOT"
for d in 0 4 8 32 128; do echo "UHC_DEBUG=$d"; UHC_DEBUG=$d python tools/probe_selfcol.py 2>/dev/null | cut -c1-200; done
echo "UHC_FAST_DENSE=40,4"; UHC_FAST_DENSE=40,4 python tools/probe_selfcol.py 2>/dev/null | cut -c1-330
