#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r05l
for cap in 9000 6000; do for nt in 1024 512; do echo "cap $cap nt $nt"; SSG_INTRO_STREAM_NT=$nt SSG_INTRO_STREAM_CAP=$cap timeout 300 python tools/time_rank_stream.py 16000 20000 2>&1 | grep N= | tee -a gpurun_out/r05l/stream_vs_lds_16k.log; done; done
