#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSSG_INTRO_PROF -I self-similarity-grouping_amd/csrc tools/micro/intro_prof.hip -o /tmp/intro_prof 2> /dev/null
/tmp/intro_prof 128000 4096 > $O/prof_128k.log 2>&1; head -14 $O/prof_128k.log
