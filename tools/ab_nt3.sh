#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/ab_nt; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "conv or embedding or fused or bottleneck or stem" 2>&1 | tail -3 | tee $O/tests3.txt
timeout 900 python tools/ab_inproc.py --reps ${REPS:-6} $VARS > $O/ab3.txt 2>&1; tail -45 $O/ab3.txt
for r in 1 2; do for pr in 0 1; do
  SSG_CONV_PAIR=$pr timeout 300 python tools/layer_table.py --reps 3 2>/dev/null | grep -E "total" | sed "s/^/PAIR=$pr /" | tee -a $O/pair3.txt
done; done
