#!/bin/bash
# Round-4 batch B: the second-generation Jaccard kernel + sparse eps / region query, range stats fix, sharded host syncs
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -k "jaccard_second or query_expansion or range_stats or rerank_stages or wide_neighbour or more_than_one or nccl or sharded_pipeline or selftraining or dropin or rerank_plain or eps or dbscan" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests_full.log 2>&1; tail -5 $O/tests_full.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-300 $O/bench.json
timeout 300 python tools/time_stages.py --reps 2 > $O/stages.txt 2>&1; tail -40 $O/stages.txt
SSG_SPARSE=0 timeout 300 python tools/time_stages.py --reps 2 > $O/stages_dense.txt 2>&1; tail -30 $O/stages_dense.txt
timeout 900 python -m pytest tests/test_dist.py -m gpu -x -q -k "bench_step" -s > $O/tests_bench_world.log 2>&1; tail -6 $O/tests_bench_world.log
