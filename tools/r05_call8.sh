#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05h; mkdir -p $O
cd $R
for v in su8 rb8k su8rb8k su8sx16; do
  echo "== $v"; SSG_LIB_PATH=$R/build_ab/libssg_$v.so timeout 600 python tools/time_rank.py 40000 128000 2>&1 | grep N= | tee -a $O/rank_$v.log
done
