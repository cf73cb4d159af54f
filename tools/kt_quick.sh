#!/bin/bash
# quick per-kernel stats of the fp32 bench step:  bash tools/kt_quick.sh <tag> [extra bench flags]
TAG=${1:-kt}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $OUT/kt.log 2>&1
python $ROOT/tools/rocpd_stats.py $OUT/kt/kt_results.db $OUT/train_step_kernel_stats.csv
rm -rf $OUT/kt
