#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06k
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
python bench.py --steps 500 --warmup 3 --no-cpu-baseline --no-roofline --no-surface 2>/dev/null | tail -1 > $OUT/bench_line_500_steps.json
python bench.py --steps 500 --warmup 3 --no-cpu-baseline --no-roofline --no-surface --config3 2>/dev/null | tail -1 > $OUT/bench_line_500_steps_config3.json
python bench.py --steps 200 --warmup 3 --no-cpu-baseline --no-roofline --no-surface --force-allreduce 2>/dev/null | tail -1 > $OUT/bench_line_200_steps_one_rank_rccl.json
python tools/train_surface_bench.py --steps 300 > $OUT/train_surface_300_steps.json 2> $OUT/err.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "honours" > $OUT/t.log 2>&1
for f in $OUT/*.json; do python - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1])
print("$f".split("/")[-1], d.get("ms_per_step") or (d.get("surface_ms_per_step"), d.get("engine_ms_per_step")), d.get("persistent_launches",{}).get("fallbacks") or d.get("counters"), d.get("loss_first_last"))
PY
done; tail -2 $OUT/t.log
