ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/inf
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -q -x -m gpu -k "lsa_step_fwd_q or infer" 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python $ROOT/tools/infer_bench.py > $OUT/kt.log 2>&1
python $ROOT/tools/rocpd_stats.py $OUT/kt/kt_results.db $OUT/infer_kernel_stats.csv >/dev/null
rm -rf $OUT/kt
python - <<PY
import csv
for r in list(csv.reader(open('$OUT/infer_kernel_stats.csv')))[:3]:
    print(r[0][:50], r[1:])
PY
for i in 1 2; do timeout 120 python $ROOT/tools/infer_bench.py 2>&1 | tail -1; done
B=32 timeout 120 python $ROOT/tools/infer_bench.py 2>&1 | tail -1
S=1000 timeout 120 python $ROOT/tools/infer_bench.py 2>&1 | tail -1
