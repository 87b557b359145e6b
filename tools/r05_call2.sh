#!/bin/bash
# round-5 GPU call 2: device-resident eps -> DBSCAN chain, streamed introsort levels (tests + timing at N = 40 k / 128 k)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "eps_rule_dbscan or sort_u64_dev or introsort or selftraining or norerank_path" > $O/tests_a.log 2>&1; tail -15 $O/tests_a.log
timeout 300 python -m pytest tests/test_gpu_chain.py -m gpu -q -x -k "config0" > $O/tests_b.log 2>&1; tail -3 $O/tests_b.log
timeout 600 python tools/time_rank.py 16000 40000 128000 > $O/rank_stream.log 2>&1; cat $O/rank_stream.log
SSG_INTRO_STREAM_NT=512 timeout 600 python tools/time_rank.py 40000 128000 > $O/rank_stream512.log 2>&1; cat $O/rank_stream512.log
SSG_INTRO_STREAM=0 timeout 600 python tools/time_rank.py 40000 128000 > $O/rank_inplace.log 2>&1; cat $O/rank_inplace.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_b.json 2> $O/bench_b.err; tail -3 $O/bench_b.err; cut -c1-300 $O/bench_b.json
