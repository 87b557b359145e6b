#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r05q
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "introsort or pairwise_vs_oracle or rerank_stages or bench_width" 2>&1 | tail -3
timeout 600 python tools/time_rank.py 20000 30000 40000 70000 128000 2>&1 | grep N= | tee gpurun_out/r05q/rank.log
timeout 300 python tools/time_stages.py --track hard --lam 0.3 --reps 3 2>&1 | grep -i "filtered1\|total" | head -4
SSG_SB_FUSED_ENC=0 timeout 300 python tools/time_stages.py --track hard --lam 0.3 --reps 3 2>&1 | grep -i "filtered1\|total" | head -4
