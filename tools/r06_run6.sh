#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06f
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
python tools/shape_stress.py 80 0 > $OUT/shape_stress_0.txt 2>&1
python tools/shape_stress.py 80 1 > $OUT/shape_stress_1.txt 2>&1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> $OUT/bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps(d.get('train_surface')))" > $OUT/bench_surface.json
tail -3 $OUT/shape_stress_*.txt; cat $OUT/bench_surface.json | cut -c1-600
