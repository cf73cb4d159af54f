#!/bin/bash
# round 6, second GPU call: the whole GPU suite under the arena (poisoned), the surface bench, bench lines after the stream / order changes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_depth.py::test_headline_shape_parity > $OUT/gpu_tests.log 2>&1
echo "gpu tests rc $?" >> $OUT/gpu_tests.log
python tools/train_surface_bench.py > $OUT/train_surface.json 2> $OUT/train_surface.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-surface > $OUT/bench_line.json 2> $OUT/bench_line.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --force-allreduce > $OUT/bench_line_one_rank_rccl.json 2>> $OUT/bench_line.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --config3 > $OUT/bench_line_config3.json 2>> $OUT/bench_line.err
cd /tmp
rocprofv3 --kernel-trace -d $OUT/ktr -o kt -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --force-allreduce > $OUT/ktr.log 2>&1
python $ROOT/tools/rocpd_timeline.py $OUT/ktr/kt_results.db $OUT/step_timeline_one_rank_rccl.txt
rm -rf $OUT/ktr
ls -la $OUT
