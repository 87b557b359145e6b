#!/bin/bash
# PMC passes over one command (kernel-trace + pmc only): usage: tools/pmc_generic.sh <tag> "<counter set 1>" "<counter set 2>" ... -- <command...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
i=0
for set in "${sets[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag/p$i -o p -- "$@" > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmc_$tag/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:60], r["Counter_Name"])
    acc[k] += float(r["Counter_Value"]); n[k] += 1
for (kn, c), v in sorted(acc.items()):
    print("%-62s %-28s %14.5g per launch (%d launches)" % (kn, c, v / n[(kn, c)], n[(kn, c)]))
PY
done
