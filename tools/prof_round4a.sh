#!/bin/bash
# Round-4 batch A (through gpurun): new / changed tests, a bench line, the input side, configs[3] / [4] with per-kernel roofline, rocprofv3
# kernel stats + one PMC traffic pass at N = 128 000.  Everything lands under gpurun_out/r04a/.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "jpeg or without_workspace or query_expansion or range_stats or conv_dual_tile or bench_width or comm_entry or rerank_stages or sharded_pipeline or nccl or dropin or selftraining or rerank_plain" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 600 python -m pytest tests -m gpu -x -q -k "conv_tile_shapes" > $O/tests_tiles.log 2>&1; tail -3 $O/tests_tiles.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-400 $O/bench.json
timeout 600 python tools/time_loader.py > $O/loader.json 2> $O/loader.err; cat $O/loader.json; tail -2 $O/loader.err
timeout 900 python tools/run_configs.py 3 4 > $O/configs.jsonl 2> $O/configs.err; cut -c1-600 $O/configs.jsonl; tail -2 $O/configs.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof128k -o cfg4 -- python $R/tools/run_configs.py 4 > $O/cfg4_under_rocprof.jsonl 2> $O/prof128k_err.log
cd $R
DB=$(find gpurun_out/r04a/prof128k -name "*results.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/r04a/kernel_stats_128k.md "rocprofv3 --kernel-trace --stats -- python tools/run_configs.py 4"
head -30 gpurun_out/r04a/kernel_stats_128k.md
timeout 900 tools/pmc_generic.sh r04_128k "FETCH_SIZE" "WRITE_SIZE" -- python $R/tools/run_configs.py 4 > $O/pmc_128k.txt 2>&1; grep -v "at::native" $O/pmc_128k.txt | cut -c1-160 | head -60
timeout 900 python -m pytest tests/test_dist.py -m gpu -x -q -k "bench_step" -s > $O/tests_bench_world.log 2>&1; tail -6 $O/tests_bench_world.log
