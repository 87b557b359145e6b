#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r05m
echo "== 2/CU nt512 cap 8500"; SSG_INTRO_STREAM_NT=512 SSG_INTRO_STREAM_CAP=8500 timeout 600 python tools/time_rank_stream.py 18000 24000 30000 36000 2>&1 | grep N= | tee -a gpurun_out/r05m/sweep.log
echo "== 40000/70000: nt512 2/CU vs default"; SSG_INTRO_STREAM_NT=512 SSG_INTRO_STREAM_CAP=8500 timeout 600 python tools/time_rank.py 40000 2>&1 | grep N= | tee -a gpurun_out/r05m/sweep.log
SSG_INTRO_STREAM_NT=512 SSG_INTRO_STREAM_CAP=6000 timeout 600 python tools/time_rank.py 70000 2>&1 | grep N= | tee -a gpurun_out/r05m/sweep.log
timeout 600 python tools/time_rank.py 70000 2>&1 | grep N= | tee -a gpurun_out/r05m/sweep.log
