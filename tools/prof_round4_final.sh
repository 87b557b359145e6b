#!/bin/bash
# Round-4 closing run (through gpurun): PMC traffic of the embedding (the source of roofline.traffic, tied to the embedding sources by
# their fingerprint), the bench line of the final build with the CPU baseline, its rocprofv3 kernel stats, the per-launch layer table.
# The whole GPU suite + smoke() run first when FULL=1.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04z; mkdir -p $O
cd $R
if [ "$FULL" = "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -4 $O/tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
fi
# SLIM=1: the embedding sources are unchanged since the last PMC pass (bench.py checks the fingerprint): bench + kernel stats + smoke only
if [ "$SLIM" = "1" ]; then
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
else
  timeout 600 bash tools/pmc_embed.sh > $O/pmc_embed.log 2>&1; tail -2 $O/pmc_embed.log; cp $R/gpurun_out/r04_pmc_conv_traffic.json $R/gpurun_out/r04_pmc_conv_traffic.md $R/profiles/
fi
timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench_final.json 2> $O/bench_final.err; tail -2 $O/bench_final.err; cut -c1-300 $O/bench_final.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/prof_err.log
cd $R
python tools/prof_summary.py $(find gpurun_out/r04z/prof -name "*results.db" | head -1) gpurun_out/r04z/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
if [ "$SLIM" != "1" ]; then timeout 600 python tools/layer_table.py > $O/layer_table.md 2>&1; tail -2 $O/layer_table.md; fi
