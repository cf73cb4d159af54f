"""Kernels of the LAST train step of a rocprofv3 rocpd trace of bench.py (Adam launch to Adam launch), grouped by (name, grid): count and
total microseconds, sorted by time.  usage: rocpd_step_list.py DB [substring]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
gcols = [c for c in ("grid_size", "grid_size_x", "grid_x") if c in cols]
q = "select %s, start, end%s from kernels order by start" % (name, (", " + gcols[0]) if gcols else "")
rows = list(db.execute(q))
adam = [i for i, r in enumerate(rows) if "adam_tf_kernel" in r[0]]
seg = rows[adam[-2] + 1: adam[-1] + 1]
agg = defaultdict(lambda: [0, 0])
for r in seg:
    key = (r[0].split("(")[0][:90], r[3] if gcols else 0)
    agg[key][0] += 1
    agg[key][1] += r[2] - r[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for (n, g), (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if sub in n:
        print("%9.1f us  %4d x  grid %-9s %s" % (ns / 1e3, c, g, n))
