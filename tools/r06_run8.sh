#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06i
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "deterministic_training or batch_uploader" > $OUT/gpu_tests.log 2>&1
echo "rc $?" >> $OUT/gpu_tests.log
MSTTS_DETERMINISTIC=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $OUT/bench_line_deterministic.json
tail -30 $OUT/gpu_tests.log | cut -c1-300; cut -c1-300 $OUT/bench_line_deterministic.json
