#!/bin/bash
# round-5 GPU call 1: the GPU tests that cover this round's host-side changes + a short bench line with the new extras
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_dist.py tests/test_abi.py -m gpu -q -x > $O/tests_dist.log 2>&1; tail -5 $O/tests_dist.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_surface.py -m gpu -q -x -k "guess_miss or sparse or range_stats or wide_reference or extract_features or embedding_vs or jpeg or loader or preprocess or selftraining" > $O/tests_sel.log 2>&1; tail -5 $O/tests_sel.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_a.json 2> $O/bench_a.err; tail -3 $O/bench_a.err; cut -c1-400 $O/bench_a.json
