#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r05o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "introsort or rerank_stages" 2>&1 | tail -3
timeout 600 python tools/time_rank.py 2000 16000 16522 18000 30000 128000 2>&1 | grep N= | tee gpurun_out/r05o/rank.log
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSSG_INTRO_PROF -I self-similarity-grouping_amd/csrc tools/micro/intro_prof.hip -o /tmp/intro_prof 2> /dev/null
/tmp/intro_prof 16000 16000 > gpurun_out/r05o/prof_16k.log 2>&1; tail -14 gpurun_out/r05o/prof_16k.log
