# L2 -> fabric traffic of the fused bottleneck kernel (separate passes, kernel-trace + pmc only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $BN_FLAGS -I $R/self-similarity-grouping_amd/csrc $R/tools/micro/bneck_prof.hip -o /tmp/bneck_plain 2>/dev/null
for set in ${PMC_SETS:-"TCC_EA0_RDREQ_sum"}; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_bneck/$tag -o p -- /tmp/bneck_plain ${1:-512} > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmc_bneck/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c, v in acc.items():
    print("%-28s %14.5g per launch (%d launches)" % (c, v / n[c], n[c]))
PY
done
