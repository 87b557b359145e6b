#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05i; mkdir -p $O
cd $R
timeout 300 python tools/layer_table.py --reps 5 > $O/lt_base.md 2>&1; tail -1 $O/lt_base.md
for v in ntA bpos1 bpos0; do
  echo "== $v"; SSG_LIB_PATH=$R/build_ab/libssg_$v.so timeout 300 python tools/layer_table.py --reps 5 > $O/lt_$v.md 2>&1; tail -1 $O/lt_$v.md
done
timeout 300 python tools/layer_table.py --reps 5 > $O/lt_base2.md 2>&1; tail -1 $O/lt_base2.md
