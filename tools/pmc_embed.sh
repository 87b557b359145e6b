# HBM traffic of the embedding's convolution launches: separate FETCH_SIZE / WRITE_SIZE passes (kernel-trace + pmc only) over
# 4 forwards of PMC_B (default 1000) images (tools/time_embed.py: warm-up + timed, original + flipped), aggregated by tools/pmc_agg.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_embed
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python $R/tools/time_embed.py --B ${PMC_B:-1000} --iters 1 > /dev/null 2>&1
  echo "== $c"; python3 $R/tools/pmc_agg.py $OUT/$c 40
done > $OUT.txt 2>&1
tail -60 $OUT.txt
python3 $R/tools/pmc_embed_summary.py $OUT.txt ${PMC_B:-1000} $R/gpurun_out/${PMC_PREFIX:-r04}_pmc_conv_traffic
