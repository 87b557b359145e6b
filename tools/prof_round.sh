set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r2a
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2a -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r2a/bench_line.json 2> $R/gpurun_out/prof_r2a/err.log
ls $R/gpurun_out/prof_r2a | head
cd $R
DB=$(find gpurun_out/prof_r2a -name "*results.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/prof_r2a/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
head -30 gpurun_out/prof_r2a/kernel_stats.md
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_r02_b.json 2> gpurun_out/bench_r02_b.err; tail -2 gpurun_out/bench_r02_b.err
python tools/run_configs.py 2 3 4 > gpurun_out/configs_r02.jsonl 2> gpurun_out/configs_r02.err; cat gpurun_out/configs_r02.jsonl | cut -c1-400
