#!/bin/bash
# fabric reads of the fused bottleneck kernel with / without the nt policy of its residual re-read and output stores
R=$GRAFT_REPO_ROOT; cd $R
for f in "" "-DSSG_BN_NT_STORE=0 -DSSG_BN_NT_RES=0"; do
  echo "## BN_FLAGS=[$f]"
  BN_FLAGS="$f" PMC_SETS="FETCH_SIZE WRITE_SIZE" bash tools/pmc_bneck.sh 1000 2>&1 | tail -8
  rm -rf $R/gpurun_out/pmc_bneck
done
