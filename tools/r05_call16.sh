#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for t in 1024 2048 3072 4096 6144; do echo "tailn $t"; SSG_INTRO_TAILN=$t timeout 300 python tools/time_rank.py 16000 2>&1 | grep N=; done
