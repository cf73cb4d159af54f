#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06u
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_persist.py tests/test_gpu_depth.py tests/test_gpu_dist.py tests/test_gpu_model.py -m gpu -q > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log | cut -c1-200
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-surface 2>/dev/null | tail -1 > $OUT/bench_line.json
python bench.py --steps 500 --warmup 3 --no-cpu-baseline --no-roofline --no-surface 2>/dev/null | tail -1 > $OUT/bench_line_500_steps.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-surface --config3 2>/dev/null | tail -1 > $OUT/bench_line_config3.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-surface --tokens 160 2>/dev/null | tail -1 > $OUT/bench_line_160_tokens.json
for f in bench_line bench_line_500_steps bench_line_config3 bench_line_160_tokens; do python - <<PY
import json
d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
print("$f", round(d["ms_per_step"],3), d["persistent_launches"]["fallbacks"], (d.get("bptt_persistent") or {}).get("frame_us"), (d.get("roofline") or {}).get("frame_us"))
PY
done
