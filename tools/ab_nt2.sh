#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/ab_nt; mkdir -p $O
timeout 900 python tools/ab_inproc.py --reps ${REPS:-6} base=self-similarity-grouping_amd/libssg_hip.so $VARS > $O/ab2.txt 2>&1; tail -45 $O/ab2.txt
