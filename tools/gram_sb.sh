#!/bin/bash
# super-block edge sweep of the int8 Gram (development aid): kernel time via rocprofv3 for SSG_I8_SB = 0 (row-major), 6, 8, 10, 12, 16
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for sb in 0 6 8 10 12 16 24; do
  SSG_I8_SB=$sb rocprofv3 --kernel-trace --stats -d $R/gpurun_out/gsb/sb$sb -o st -- python $R/tools/time_stages.py --track hard --lam 0.3 --reps 2 > /dev/null 2>&1
  python3 - $R/gpurun_out/gsb/sb$sb $sb <<'PY'
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0]
c = sqlite3.connect(db)
for name, n, avg in c.execute("select name, count(*), avg(duration) from kernels where name like '%gram_i8_kernel%' group by name"):
    print("SSG_I8_SB=%s gram_i8_kernel: %d launches, avg %.1f us" % (sys.argv[2], n, avg / 1e3))
PY
done
SSG_I8_SB=10 $R/tools/pmc_generic.sh gsb10 "FETCH_SIZE" -- python $R/tools/time_stages.py --track hard --lam 0.3 --reps 1 2>&1 | grep gram_i8_kernel
