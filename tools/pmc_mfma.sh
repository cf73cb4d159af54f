#!/bin/bash
# MFMA-busy counters per kernel: separate --pmc passes with --kernel-trace only (no other trace domains), short workload.
cd /tmp && export TMPDIR=/tmp
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES MfmaUtil; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmcm_$c -o pmc -- python /root/repo/bench.py --steps 1 --warmup 1 --frames 48 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
cd /root/repo
python - <<'PY'
import csv, glob, re, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in glob.glob("gpurun_out/pmcm_*"):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r.get("Kernel_Name", "")).replace("void ", "").replace("mstts::", "")
            a = acc[name][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
cs = sorted({c for v in acc.values() for c in v})
with open("gpurun_out/pmc_mfma_summary.csv", "w", newline="") as fh:
    w = csv.writer(fh); w.writerow(["kernel", "calls"] + ["avg_" + c for c in cs] + ["mfma_busy_over_sq_busy"])
    rows = []
    for k, v in acc.items():
        calls = max(x[1] for x in v.values())
        avg = {c: (v[c][0] / v[c][1] if c in v and v[c][1] else 0.0) for c in cs}
        ratio = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / avg["SQ_BUSY_CYCLES"] if avg.get("SQ_BUSY_CYCLES") else 0.0
        rows.append((-avg.get("SQ_BUSY_CYCLES", 0.0) * calls, [k, calls] + ["%.1f" % avg[c] for c in cs] + ["%.4f" % ratio]))
    for _, r in sorted(rows): w.writerow(r)
print(open("gpurun_out/pmc_mfma_summary.csv").read()[:3000])
PY
rm -rf gpurun_out/pmcm_*
