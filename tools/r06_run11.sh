#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06m
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_persist.py -m gpu -q -k "surface or variable_length or cli_to_training or uploader or abort or equals_launch" > $OUT/t.log 2>&1
tail -3 $OUT/t.log
python tools/train_surface_bench.py --steps 100 > $OUT/train_surface_100_steps.json 2> $OUT/err.txt
python tools/train_surface_bench.py --steps 50 > $OUT/train_surface_50_steps.json 2>> $OUT/err.txt
for f in $OUT/*.json; do python - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1])
print("$f".split("/")[-1], d.get("surface_ms_per_step"), d.get("engine_ms_per_step"), d.get("surface_over_engine"), d.get("surface_host_ms_per_step"), d.get("counters"))
PY
done
