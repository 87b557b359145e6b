#!/bin/bash
# Round-4 final measurement batch (through gpurun): the whole GPU suite, the bench line (with the CPU baseline), rocprofv3 kernel stats of
# the bench, PMC traffic of the embedding (the source of roofline.traffic), BASELINE configs [0]-[4] with per-kernel roofline objects at
# configs [3] / [4], the input side, the per-launch layer table, PMC traffic of the grouping kernels.  Everything lands under gpurun_out/r04/;
# the summaries are copied to profiles/ by hand.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -6 $O/tests.log
timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench_final.json 2> $O/bench_final.err; tail -2 $O/bench_final.err; cut -c1-300 $O/bench_final.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/prof_err.log
cd $R
DB=$(find gpurun_out/r04/prof -name "*results.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/r04/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
head -14 gpurun_out/r04/kernel_stats.md | cut -c1-160
timeout 600 bash tools/pmc_embed.sh > $O/pmc_embed.log 2>&1; tail -3 $O/pmc_embed.log
timeout 900 python tools/run_configs.py 0 1 2 3 4 > $O/configs.jsonl 2> $O/configs.err; cut -c1-250 $O/configs.jsonl; tail -2 $O/configs.err
timeout 600 python tools/time_loader.py > $O/loader.json 2> $O/loader.err; cat $O/loader.json
timeout 600 python tools/layer_table.py > $O/layer_table.md 2>&1; tail -3 $O/layer_table.md
timeout 900 tools/pmc_generic.sh r04_group "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" -- python $R/tools/time_stages.py --track hard --lam 0.3 --reps 1 > $O/pmc_group.txt 2>&1; grep -v "at::native" $O/pmc_group.txt | grep -i "jaccard\|region\|compact\|gram\|introsort\|sbound" | cut -c1-150 | head -30
