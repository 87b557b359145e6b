#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for n in 16000 30000; do for v in "" _ab/gi_nt.so _ab/jac_nt.so "" _ab/gi_nt.so _ab/jac_nt.so; do
  echo "## N=$n lib=[$v]"
  SSG_LIB_PATH=${v:+$R/$v} timeout 300 python tools/time_stages.py --N $n --Ns 12936 --track hard --lam 0.3 --reps 3 2>&1 | grep -i "sqdist_self\|jaccard_rows\|introsort\|total\|eps_compact" | head -8
done; done
