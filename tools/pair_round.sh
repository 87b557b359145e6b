#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/ab_nt; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "conv_pair" 2>&1 | tail -2
VARIANTS=';-DSSG_PAIR_ABL_NOEPI' bash tools/pair_prof.sh
for r in 1 2; do for pr in 0 1; do
  SSG_CONV_PAIR=$pr timeout 300 python tools/layer_table.py --reps 3 2>/dev/null | grep -E "pair conv3|total" | sed "s/^/PAIR=$pr /"
done; done
