#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r05k
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSSG_INTRO_PROF -I self-similarity-grouping_amd/csrc tools/micro/intro_prof.hip -o /tmp/intro_prof 2> /dev/null
/tmp/intro_prof 16000 16000 > gpurun_out/r05k/prof_16k.log 2>&1; tail -17 gpurun_out/r05k/prof_16k.log
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I self-similarity-grouping_amd/csrc tools/micro/intro_prof.hip -o /tmp/intro_noprof 2> /dev/null
/tmp/intro_noprof 16000 16000 | tail -1
