#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "introsort" > $O/tests_a.log 2>&1; tail -4 $O/tests_a.log
timeout 600 python tools/time_rank.py 40000 128000 > $O/rank_a.log 2>&1; grep N= $O/rank_a.log
SSG_INTRO_STREAM_NT=512 timeout 600 python tools/time_rank.py 40000 128000 > $O/rank_b.log 2>&1; grep N= $O/rank_b.log
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSSG_INTRO_PROF -I self-similarity-grouping_amd/csrc tools/micro/intro_prof.hip -o /tmp/intro_prof 2> /dev/null
/tmp/intro_prof 128000 4096 > $O/prof_128k.log 2>&1; head -14 $O/prof_128k.log
