#!/bin/bash
# round-5 GPU call 4: cycle accounting of the streamed introsort kernel
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSSG_INTRO_PROF -I self-similarity-grouping_amd/csrc tools/micro/intro_prof.hip -o /tmp/intro_prof 2> /dev/null
/tmp/intro_prof 128000 4096 > $O/prof_128k.log 2>&1; tail -12 $O/prof_128k.log
/tmp/intro_prof 40000 4096 > $O/prof_40k.log 2>&1; tail -12 $O/prof_40k.log
SSG_INTRO_STREAM_NT=512 /tmp/intro_prof 128000 4096 > $O/prof_128k_512.log 2>&1; tail -12 $O/prof_128k_512.log
