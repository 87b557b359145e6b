# LDS / issue counters of the convolution kernels on one layer shape (kernel-trace + pmc only): tools/pmc_conv.sh c3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${1:-c3}
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  MICRO_B=512 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_conv/$tag -o p -- python $R/tools/conv_micro.py $CFG split > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmc_conv/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:48]
    if "conv_" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    for c, v in d.items():
        print("%-50s %-30s %14.5g per launch (%d)" % (k, c, v / n[(k, c)], n[(k, c)]))
PY
done
