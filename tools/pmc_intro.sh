# PMC counters of the introsort ranking kernel (separate passes, kernel-trace + pmc only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_intro/$tag -o p -- python $R/tools/time_rank.py 16000 > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmc_intro/$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    if "introsort" not in k and "topk_rank_kernel" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    for c, v in d.items():
        print("%-62s %-24s %14.4g per launch (%d launches)" % (k, c, v / n[(k, c)], n[(k, c)]))
PY
done
