#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for f in "" "-DSSG_BN_ABL_NORES" "-DBN_LAYER2" "-DBN_LAYER2 -DSSG_BN_ABL_NORES"; do
  echo "## flags [$f]"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSSG_BN_PROF $f -I self-similarity-grouping_amd/csrc tools/micro/bneck_prof.hip -o /tmp/bneck_prof 2>/dev/null && /tmp/bneck_prof 1000
done
