set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r2f
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2f -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r2f/bench_line.json 2> $R/gpurun_out/prof_r2f/err.log
ls $R/gpurun_out/prof_r2f | head
cd $R
DB=$(find gpurun_out/prof_r2f -name "*results.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/prof_r2f/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
head -30 gpurun_out/prof_r2f/kernel_stats.md
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_r02_f.json 2> gpurun_out/bench_r02_f.err; tail -2 gpurun_out/bench_r02_f.err
python tools/run_configs.py 2 3 4 > gpurun_out/configs_r02f.jsonl 2> gpurun_out/configs_r02f.err; cat gpurun_out/configs_r02f.jsonl | cut -c1-400
