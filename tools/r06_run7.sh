#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06h
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8 9 10; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>> $OUT/bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); t = d.get('train_surface') or {}
print('in bench', $i, t.get('error') or (t.get('surface_ms_per_step'), t.get('loss_first_last'), t.get('counters')))" >> $OUT/surface_runs.txt
done
for i in 1; do
python tools/train_surface_bench.py 2>> $OUT/bench.err | python -c "
import sys, json
t = json.loads(sys.stdin.readline())
print('standalone', $i, (t.get('surface_ms_per_step'), t.get('loss_first_last'), t.get('counters')))" >> $OUT/surface_runs.txt 2>&1
done
cat $OUT/surface_runs.txt
