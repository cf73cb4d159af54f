"""Where the GPU idles inside a train step: from a rocprofv3 rocpd database of `bench.py`, the last step (Adam launch to Adam launch),
the union of kernel time, the idle time, and the idle time grouped by the kernel that FOLLOWS the gap.
usage: rocpd_gaps.py DB [OUT.txt]"""
import sqlite3
import sys
from collections import defaultdict


def main(db_path, out):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = list(db.execute("select %s, start, end from kernels order by start" % name))
    adam = [i for i, r in enumerate(rows) if "adam_tf_kernel" in r[0]]
    if len(adam) < 2:
        print("fewer than two optimizer steps in the trace", file=out)
        return
    seg = rows[adam[-2] + 1: adam[-1] + 1]
    t0, t1 = rows[adam[-2]][2], seg[-1][2]
    busy, cur_end, gaps = 0, t0, defaultdict(lambda: [0, 0])
    for n, s, e in seg:
        if s > cur_end:
            g = gaps[n.split("(")[0][:70]]
            g[0] += s - cur_end
            g[1] += 1
        busy += max(0, e - max(s, cur_end))
        cur_end = max(cur_end, e)
    wall = t1 - t0
    print("step (end of Adam to end of Adam): %.3f ms, kernels busy %.3f ms, idle %.3f ms (%.1f %%), %d launches" %
          (wall / 1e6, busy / 1e6, (wall - busy) / 1e6, 100.0 * (wall - busy) / wall, len(seg)), file=out)
    print("idle time by the kernel that follows the gap:", file=out)
    for k, (ns, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print("  %9.1f us in %5d gaps (%.2f us each)  %s" % (ns / 1e3, c, ns / 1e3 / c, k), file=out)


if __name__ == "__main__":
    main(sys.argv[1], open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
