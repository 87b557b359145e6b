#!/bin/bash
# Round-3 measurement batch (run through gpurun): final bench line, rocprofv3 kernel stats, PMC traffic + MFMA-busy passes, configs, layer table.
# Everything lands under gpurun_out/r03/; the summaries are copied to profiles/ by hand.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python bench.py --steps 2 --warmup 1 > $O/bench_final.json 2> $O/bench_final.err; tail -2 $O/bench_final.err
python tools/layer_table.py > $O/layer_table.md 2>&1
python tools/run_configs.py 0 1 2 3 4 > $O/configs.jsonl 2> $O/configs.err; cut -c1-300 $O/configs.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/prof_err.log
cd $R
DB=$(find gpurun_out/r03/prof -name "*results.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/r03/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
head -16 gpurun_out/r03/kernel_stats.md
bash tools/pmc_embed.sh > $O/pmc_embed.log 2>&1; tail -3 $O/pmc_embed.log
# MFMA-busy / wave-state counters per kernel family: the embedding and the grouping leg
tools/pmc_generic.sh r03_embed "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" -- python $R/tools/time_embed.py --B 1000 --iters 1 > $O/pmc_mfma_embed.txt 2>&1
tools/pmc_generic.sh r03_group "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" -- python $R/tools/time_stages.py --reps 1 > $O/pmc_group.txt 2>&1
grep -E "conv_dma|bottleneck|stem|gram_i8|sbound" $O/pmc_mfma_embed.txt $O/pmc_group.txt | grep -E "MFMA_BUSY|GUI_ACTIVE" | cut -c1-200 | head -30
