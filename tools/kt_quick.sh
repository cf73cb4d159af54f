#!/bin/bash
# kernel trace of the headline bench command -> gpurun_out/<tag>_train_step_kernel_stats.csv  (bash tools/kt_quick.sh r03)
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/kt_$TAG -o kt -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/kt_$TAG.log 2>&1
python $ROOT/tools/rocpd_stats.py $OUT/kt_$TAG/kt_results.db $OUT/${TAG}_train_step_kernel_stats.csv
python $ROOT/tools/rocpd_gaps.py $OUT/kt_$TAG/kt_results.db $OUT/${TAG}_train_step_gaps.txt
cat $OUT/${TAG}_train_step_gaps.txt
rm -rf $OUT/kt_$TAG
head -40 $OUT/${TAG}_train_step_kernel_stats.csv
