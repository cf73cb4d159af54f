"""Aggregate rocprofv3 --pmc counter_collection CSVs (one pass per counter) into per-kernel per-launch averages.
usage: pmc_summary.py OUT.csv PASS_DIR [PASS_DIR ...]   (FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB)"""
import csv, glob, os, re, sys
from collections import defaultdict
out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))      # kernel -> counter -> [sum, n]
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r.get("Kernel_Name", "")).replace("void ", "").replace("mstts::", "")
            if not name:
                continue
            key = (name, r.get("Grid_Size", ""))
            a = acc[key][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
counters = sorted({c for v in acc.values() for c in v})
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    hdr = ["kernel", "grid_size", "calls"]
    for c in counters:
        hdr.append("avg_%s%s" % (c, "_KB" if c in ("FETCH_SIZE", "WRITE_SIZE") else ""))
        if c == "FETCH_SIZE":
            hdr.append("avg_FETCH_SIZE_KB_x2_gfx950_correction")
    w.writerow(hdr)
    rows = []
    for (name, grid), cs in acc.items():
        calls = max(v[1] for v in cs.values())
        row = [name, grid, calls]
        for c in counters:
            avg = cs[c][0] / cs[c][1] if c in cs and cs[c][1] else 0.0
            row.append("%.2f" % avg)
            if c == "FETCH_SIZE":
                row.append("%.2f" % (2 * avg))
        rows.append((-(float(row[3]) * calls), row))
    for _, row in sorted(rows):
        w.writerow(row)
print("wrote", out, len(acc), "kernels")
