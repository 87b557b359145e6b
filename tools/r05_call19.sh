#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r05r
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "self_distance or rerank_stages or bench_width or reranking_dropin or norerank or wide_neighbourhoods" 2>&1 | tail -4
timeout 300 python tools/time_stages.py --track hard --lam 0.3 --reps 3 2>&1 | grep -i "sqdist_self_i8\|gram_i8_encode "
SSG_I8_DMA=0 timeout 300 python tools/time_stages.py --track hard --lam 0.3 --reps 3 2>&1 | grep -i "sqdist_self_i8\|gram_i8_encode "
