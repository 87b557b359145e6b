#!/bin/bash
# Round-5 measurement batch (through gpurun): PMC traffic of the embedding (source of roofline.traffic, tied to the embedding sources by
# their fingerprint), the bench line with the CPU baseline, rocprofv3 kernel stats of the same command -- product default (two streams:
# durations overlap) AND single stream (SSG_FLIP_STREAMS=0: every duration is a launch running alone) --, the per-launch layer table,
# configs[0..4].  PARTS selects: pmc bench prof1 prof2 layer configs (default all)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05z; mkdir -p $O
cd $R
PARTS=${PARTS:-"pmc bench prof2 prof1 layer configs pmcgroup"}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has pmc; then PMC_PREFIX=r05 timeout 900 bash tools/pmc_embed.sh > $O/pmc_embed.log 2>&1; tail -2 $O/pmc_embed.log; cp $R/gpurun_out/r05_pmc_conv_traffic.json $R/gpurun_out/r05_pmc_conv_traffic.md $R/profiles/; fi
if has bench; then timeout 1200 python bench.py --steps 5 --warmup 2 > $O/bench_final.json 2> $O/bench_final.err; tail -2 $O/bench_final.err; cut -c1-300 $O/bench_final.json; fi
cd /tmp && export TMPDIR=/tmp
if has prof2; then
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof2 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof_two_streams.json 2> $O/prof2_err.log
  (cd $R; python tools/prof_summary.py $(find gpurun_out/r05z/prof2 -name "*results.db" | head -1) gpurun_out/r05z/kernel_stats_two_streams.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras" "PRODUCT DEFAULT: the two forwards of a batch (original + flipped) run on two HIP streams, so the durations below OVERLAP -- their sum is about twice the wall time of the step and every convolution average is dilated by the kernel running beside it (about 1.6x).  Per-kernel fractions must be taken from the single-stream file next to this one (r05_bench_kernel_stats_single_stream.md) or from r05_layer_table.md.")
fi
if has prof1; then
  SSG_RERANK_OVERLAP=0 SSG_FLIP_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof_single_stream.json 2> $O/prof1_err.log
  (cd $R; python tools/prof_summary.py $(find gpurun_out/r05z/prof1 -name "*results.db" | head -1) gpurun_out/r05z/kernel_stats_single_stream.md "SSG_RERANK_OVERLAP=0 SSG_FLIP_STREAMS=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras" "SINGLE STREAM (SSG_FLIP_STREAMS=0): every launch runs alone, durations do not overlap and sum to the step's GPU time; this is the file per-kernel averages and fractions are quoted from.  The product default (two streams, +2-3 % throughput) is profiled in r05_bench_kernel_stats_two_streams.md.")
fi
cd $R
if has layer; then timeout 600 python tools/layer_table.py --reps 5 > $O/layer_table.md 2>&1; tail -2 $O/layer_table.md; fi
if has configs; then timeout 1500 python tools/run_configs.py > $O/configs.jsonl 2> $O/configs.err; cut -c1-400 $O/configs.jsonl; tail -2 $O/configs.err; fi
if has pmcgroup; then
  # HBM traffic of the grouping kernels (FETCH_SIZE counts half of wide coalesced reads on gfx950: see MI355X_MICROARCH.md) at the bench size and at N = 128 000
  timeout 600 tools/pmc_generic.sh r05_group "FETCH_SIZE" "WRITE_SIZE" -- python $R/tools/time_stages.py --track hard --lam 0.3 --reps 1 > $O/pmc_group_hard.txt 2>&1
  grep -v "at::native" $O/pmc_group_hard.txt | grep -i "jaccard\|region\|compact\|gram_i8_kernel\|introsort\|sbound\|bitonic" | cut -c1-150 | head -30
  timeout 900 tools/pmc_generic.sh r05_n128k "FETCH_SIZE" "WRITE_SIZE" -- python $R/tools/time_rank.py 128000 > $O/pmc_rank_n128k.txt 2>&1
  grep -i "introsort" $O/pmc_rank_n128k.txt | cut -c1-150
fi
