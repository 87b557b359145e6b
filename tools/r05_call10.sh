#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r05j
for cap in 12000 16000 20000; do SSG_INTRO_STREAM_CAP=$cap timeout 300 python tools/time_rank_stream.py 20000 24000 30000 36000 2>&1 | grep N= | tee -a gpurun_out/r05j/stream_vs_lds.log; done
