#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r05n
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "introsort" 2>&1 | tail -3
timeout 600 python tools/time_rank.py 16000 18000 20000 30000 40000 70000 128000 2>&1 | grep N= | tee gpurun_out/r05n/rank_policy.log
