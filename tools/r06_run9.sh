#!/bin/bash
# full GPU suite on the final tree (incl. the headline-shape parity test) + a fresh headline line
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06j
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1
echo "gpu tests rc $?" >> $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
tail -15 $OUT/gpu_tests.log | cut -c1-250; cat $OUT/smoke.log | tail -2
