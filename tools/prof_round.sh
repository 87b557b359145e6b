set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r2e
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2e -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r2e/bench_line.json 2> $R/gpurun_out/prof_r2e/err.log
ls $R/gpurun_out/prof_r2e | head
cd $R
DB=$(find gpurun_out/prof_r2e -name "*results.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/prof_r2e/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
head -30 gpurun_out/prof_r2e/kernel_stats.md
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_r02_e.json 2> gpurun_out/bench_r02_e.err; tail -2 gpurun_out/bench_r02_e.err
python tools/run_configs.py 2 3 4 > gpurun_out/configs_r02e.jsonl 2> gpurun_out/configs_r02e.err; cat gpurun_out/configs_r02e.jsonl | cut -c1-400
bash tools/pmc_embed.sh > gpurun_out/pmc_embed.log 2>&1; tail -30 gpurun_out/pmc_embed.txt
