#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result into the per-kernel stats table committed under profiles/.
usage: tools/prof_summary.py gpurun_out/<dir>/<name>_results.db profiles/<name>.md "<command that was profiled>" """
import sqlite3
import sys


def main():
    db, out, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc"))
    unit = "ns"   # rocpd `kernels.duration` = end - start in nanoseconds
    tot = sum(r[2] for r in rows)
    rows = [(r[0], r[1], r[2], r[3], 100.0 * r[2] / tot) for r in rows]
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary\n\n")
        f.write("command: `%s`\n\nsource: %s (view `kernels`, grouped by kernel name), durations in %s; total kernel time %.3f ms\n\n" % (cmd, db, unit, tot / 1e6))
        if note:
            f.write("**%s**\n\n" % note)
        f.write("| kernel | calls | total (ms) | avg (us) | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows:
            if r[4] < 0.001 and r[1] < 2:
                continue
            f.write("| `%s` | %d | %.3f | %.2f | %.2f |\n" % (r[0][:110], r[1], r[2] / 1e6, r[3] / 1e3, r[4]))
    print("wrote", out)


if __name__ == "__main__":
    main()
