#!/bin/bash
# phase stamps + ablations of conv_pair_kernel (tools/micro/pair_prof.hip) on the GPU box -> gpurun_out/pair_prof.txt
# VARIANTS: ';'-separated lists of -D switches (default: shipped kernel + the ablations)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/pair_prof.txt; : > $O
VARIANTS=${VARIANTS:-";-DSSG_PAIR_ABL_NOEPI;-DSSG_PAIR_ABL_NOMMA;-DSSG_PAIR_ABL_NOEPI -DSSG_PAIR_ABL_NOMMA;-DSSG_PAIR_ABL_NORES;-DSSG_PAIR_ABL_NOSTORE;-DSSG_PAIR_AUX=2"}
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  echo "## build switches: [$v]" >> $O
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSSG_PAIR_PROF $v -I self-similarity-grouping_amd/csrc -I include tools/micro/pair_prof.hip -o /tmp/pair_prof 2>>$O.err && timeout 120 /tmp/pair_prof >> $O 2>&1
done
cat $O
