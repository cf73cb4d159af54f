"""Every kernel of the LAST train step of a rocprofv3 rocpd trace of bench.py (Adam launch to Adam launch) in start order: offset from the
step's start, duration, how much of it overlaps the kernel in front (another stream), queue / stream ids when the trace has them, grid, name.
usage: rocpd_timeline.py DB [OUT.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
extra = [c for c in ("queue_id", "stream_id", "grid_size", "grid_size_x", "grid_x", "workgroup_size", "workgroup_size_x") if c in cols]
rows = list(db.execute("select %s, start, end%s from kernels order by start" % (name, "".join(", " + c for c in extra))))
adam = [i for i, r in enumerate(rows) if "adam_tf_kernel" in r[0]]
seg = rows[adam[-2] + 1: adam[-1] + 1]
t0 = rows[adam[-2]][2]
print("# offset_us  dur_us  overlap_with_earlier_us  %s  name" % "  ".join(extra), file=out)
latest_end = t0
for r in seg:
    n, s, e = r[0], r[1], r[2]
    ov = max(0, min(e, latest_end) - s)
    print("%10.1f %9.1f %9.1f  %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, ov / 1e3, "  ".join(str(x) for x in r[3:]), n.split("(")[0][:110]), file=out)
    latest_end = max(latest_end, e)
print("# step %.3f ms, %d kernels" % ((seg[-1][2] - t0) / 1e6, len(seg)), file=out)
