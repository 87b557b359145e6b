#!/bin/bash
# Round-6 measurement batch (through gpurun).  PARTS selects (default all):
#   pmc      PMC traffic of the embedding (source of roofline.traffic, tied to the embedding sources by their fingerprint)
#   bench    the bench line with the CPU baseline + the --uniform cross-check line
#   prof1/2  rocprofv3 kernel stats of the bench command: single stream (every duration a launch running alone) / product default
#   layer    per-launch layer table
#   configs  BASELINE configs[0..4]
#   pmcchain FETCH_SIZE / WRITE_SIZE / MFMA-busy passes over the grouping leg AS TIMED (tools/time_group_chain.py: the device chain) at
#            N = 16 000 and, for the streamed introsort + the big sample sort, at N = 128 000
#   micro    the development probes quoted in DESIGN section 11 (A/B of the 8-wave conv tiles, dense passes, sorts, eps chain)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06z; mkdir -p $O
cd $R
PARTS=${PARTS:-"pmc bench prof1 prof2 layer configs pmcchain micro"}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has pmc; then PMC_PREFIX=r06 timeout 900 bash tools/pmc_embed.sh > $O/pmc_embed.log 2>&1; tail -2 $O/pmc_embed.log; cp $R/gpurun_out/r06_pmc_conv_traffic.json $R/gpurun_out/r06_pmc_conv_traffic.md $R/profiles/; fi
if has bench; then
  timeout 1200 python bench.py --steps 5 --warmup 2 > $O/bench_final.json 2> $O/bench_final.err; tail -2 $O/bench_final.err; cut -c1-300 $O/bench_final.json
  timeout 600 python bench.py --steps 5 --warmup 2 --uniform --no-cpu-baseline --no-extras > $O/bench_uniform.json 2> $O/bench_uniform.err; cut -c1-200 $O/bench_uniform.json
fi
cd /tmp && export TMPDIR=/tmp
if has prof2; then
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof2 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof_two_streams.json 2> $O/prof2_err.log
  (cd $R; python tools/prof_summary.py $(find gpurun_out/r06z/prof2 -name "*results.db" | head -1) gpurun_out/r06z/kernel_stats_two_streams.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras" "PRODUCT DEFAULT: the two forwards of a batch (original + flipped) run on two HIP streams, so the durations below OVERLAP -- their sum is about twice the wall time of the step and every convolution average is dilated by the kernel running beside it (about 1.6x).  Per-kernel fractions must be taken from the single-stream file next to this one (r06_bench_kernel_stats_single_stream.md) or from r06_layer_table.md.")
fi
if has prof1; then
  SSG_RERANK_OVERLAP=0 SSG_FLIP_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof_single_stream.json 2> $O/prof1_err.log
  (cd $R; python tools/prof_summary.py $(find gpurun_out/r06z/prof1 -name "*results.db" | head -1) gpurun_out/r06z/kernel_stats_single_stream.md "SSG_RERANK_OVERLAP=0 SSG_FLIP_STREAMS=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras" "SINGLE STREAM (SSG_FLIP_STREAMS=0, SSG_RERANK_OVERLAP=0): every launch runs alone, durations do not overlap and sum to the step's GPU time; this is the file per-kernel averages and fractions are quoted from.  The product default (two streams, +2-3 % throughput) is profiled in r06_bench_kernel_stats_two_streams.md.")
fi
cd $R
if has layer; then timeout 600 python tools/layer_table.py --reps 5 > $O/layer_table.md 2>&1; tail -2 $O/layer_table.md; fi
if has configs; then timeout 1500 python tools/run_configs.py > $O/configs.jsonl 2> $O/configs.err; cut -c1-300 $O/configs.jsonl; tail -2 $O/configs.err; fi
if has pmcchain; then
  C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
  timeout 600 tools/pmc_generic.sh r06_chain "FETCH_SIZE" "WRITE_SIZE" "$C" -- python $R/tools/time_group_chain.py --reps 1 > $O/pmc_chain_n16k.txt 2>&1
  grep -v "at::native" $O/pmc_chain_n16k.txt | cut -c1-150 | head -80
  timeout 900 tools/pmc_generic.sh r06_chain128 "FETCH_SIZE" "WRITE_SIZE" "$C" -- python $R/tools/time_group_chain.py --N 128000 --Ns 4000 --d 256 --lam 0.1 --track separable --reps 1 > $O/pmc_chain_n128k.txt 2>&1
  grep -v "at::native" $O/pmc_chain_n128k.txt | grep -i "introsort\|ss_\|compact\|gram\|jaccard" | cut -c1-150 | head -40
  timeout 300 tools/pmc_generic.sh r06_embed "$C" -- python $R/tools/time_embed.py --B 1000 --iters 1 > $O/pmc_mfma_embed.txt 2>&1
  python3 tools/pmc_mfma_summary.py $O/pmc_mfma_embed.txt $O/pmc_chain_n16k.txt > $O/pmc_mfma_table.md; cut -c1-170 $O/pmc_mfma_table.md
fi
if has micro; then
  timeout 600 python tools/ab_inproc.py --reps 8 base=self-similarity-grouping_amd/libssg_hip.so wm128=self-similarity-grouping_amd/libssg_hip.so,SSG_CONV_TALL_WM=128 wm128x2=self-similarity-grouping_amd/libssg_hip.so,SSG_CONV_TALL_WM=129 > $O/ab_conv_wm.txt 2>&1; tail -4 $O/ab_conv_wm.txt
  (SSG_TC_NOCAND=1 timeout 300 python tools/time_compact.py 16000 30000 64000 2>&1 | grep "N=") > $O/dense_passes.txt; cat $O/dense_passes.txt
  (timeout 300 python tools/time_sort.py 270000 940000 4000000 17000000 2>&1 | grep "n=") > $O/sorts.txt; cat $O/sorts.txt
  (for k in 1 0; do SSG_EPS_FUSED_LAUNCHES=$k timeout 200 python tools/time_eps_chain.py 2>&1 | tail -1 | sed "s/^/SSG_EPS_FUSED_LAUNCHES=$k: /"; done) > $O/eps_chain_timing.txt; cat $O/eps_chain_timing.txt
fi
