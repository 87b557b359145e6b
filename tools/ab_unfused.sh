#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for r in 1 2; do for f in 1 0; do
  echo "## SSG_FUSED_BOTTLENECK=$f"
  SSG_FUSED_BOTTLENECK=$f timeout 300 python tools/layer_table.py --reps 3 2>/dev/null | grep -E "^\| +([0-9]|1[0-9]|2[0-2]) \||total" | cut -c1-110
done; done
