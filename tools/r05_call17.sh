#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p; mkdir -p $O
cd $R
timeout 300 python tools/layer_table.py --reps 5 2>&1 | tail -1
for v in ilp memclause bias0; do
  echo "== $v"; SSG_LIB_PATH=$R/build_ab/libssg_$v.so timeout 300 python tools/layer_table.py --reps 5 2>&1 | tail -1
  SSG_LIB_PATH=$R/build_ab/libssg_$v.so timeout 300 python tools/time_stages.py --track hard --lam 0.3 --reps 3 2>&1 | grep -i "total\|introsort\|sqdist_self_i8\|filtered1\|jaccard_rows2" | head -8
done
echo "== base stages"; timeout 300 python tools/time_stages.py --track hard --lam 0.3 --reps 3 2>&1 | grep -i "total\|introsort\|sqdist_self_i8\|filtered1\|jaccard_rows2" | head -8
