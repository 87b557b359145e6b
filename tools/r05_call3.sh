#!/bin/bash
# round-5 GPU call 3: streamed introsort with batched loads, workgroups per CU / thread-count sweep
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "introsort" > $O/tests_a.log 2>&1; tail -4 $O/tests_a.log
timeout 600 python tools/time_rank.py 40000 128000 > $O/rank_a.log 2>&1; grep N= $O/rank_a.log
SSG_INTRO_STREAM_CAP=4500 timeout 600 python tools/time_rank.py 40000 128000 > $O/rank_b.log 2>&1; grep N= $O/rank_b.log
SSG_INTRO_STREAM_CAP=4500 SSG_INTRO_STREAM_NT=512 timeout 600 python tools/time_rank.py 40000 128000 > $O/rank_c.log 2>&1; grep N= $O/rank_c.log
SSG_INTRO_STREAM_CAP=12000 timeout 600 python tools/time_rank.py 40000 > $O/rank_d.log 2>&1; grep N= $O/rank_d.log
SSG_INTRO_STREAM_CAP=12000 SSG_INTRO_STREAM_NT=512 timeout 600 python tools/time_rank.py 40000 > $O/rank_e.log 2>&1; grep N= $O/rank_e.log
