#!/bin/bash
# fabric traffic of conv_pair_kernel with the default cache policy and with nt on the residual / output streams (rocprofv3 PMC, separate passes)
R=$GRAFT_REPO_ROOT; cd $R
for a in 0 2; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSSG_PAIR_AUX=$a -I self-similarity-grouping_amd/csrc -I include tools/micro/pair_prof.hip -o /tmp/pair_prof_$a 2>/dev/null
  echo "## SSG_PAIR_AUX=$a (KB per launch; FETCH_SIZE counts 64 of every 128 bytes of a wide read)"
  bash tools/pmc_generic.sh pair$a "FETCH_SIZE" "WRITE_SIZE" -- /tmp/pair_prof_$a 2>&1 | grep conv_pair
  rm -rf $R/gpurun_out/pmc_pair$a
done
