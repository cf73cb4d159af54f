"""Per-kernel statistics (calls, total, average, min, max duration) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace -d DIR -o NAME` writes DIR/NAME_results.db).  usage: rocpd_stats.py DB [OUT.csv]"""
import csv
import sqlite3
import sys


def stats(db_path):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    q = "select %s, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by %s order by 3 desc" % (name, name)
    rows = list(db.execute(q))
    tot = sum(r[2] for r in rows) or 1
    return [(r[0], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]) for r in rows]


if __name__ == "__main__":
    rows = stats(sys.argv[1])
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        out.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.2f" % r[4], r[5], r[6]])
