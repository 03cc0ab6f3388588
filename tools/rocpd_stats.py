"""Per-kernel summary of a rocprofv3 rocpd database (run_results.db): calls, average / total duration.
usage: python tools/rocpd_stats.py gpurun_out/profNN/run_results.db [steps] [out.csv]"""
import csv
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, count(*), avg(end - start), sum(end - start) from kernels group by %s order by 4 desc"
                   % (name_col, name_col)).fetchall()
total = sum(r[3] for r in rows)
out = [("Name", "Calls", "AverageNs", "TotalDurationNs", "Percentage")]
print("total kernel time %.3f ms%s" % (total / 1e6, "  (%.1f us/step)" % (total / 1e3 / steps) if steps else ""))
for name, calls, avg, tot in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"^void ", "", short)[:64]
    per = "  %7.1f us/step" % (tot / 1e3 / steps) if steps else ""
    print("%-64s calls %5d avg %8.1f us %5.1f%%%s" % (short, calls, avg / 1e3, 100.0 * tot / total, per))
    out.append((name, calls, "%.1f" % avg, tot, "%.2f" % (100.0 * tot / total)))
if len(sys.argv) > 3:
    with open(sys.argv[3], "w", newline="") as f:
        csv.writer(f).writerows(out)
