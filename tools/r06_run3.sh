#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
python tools/train_surface_bench.py > $OUT/train_surface.json 2> $OUT/train_surface.err
MSTTS_FEEDER_WORKERS=0 python tools/train_surface_bench.py > $OUT/train_surface_no_workers.json 2>> $OUT/train_surface.err
MSTTS_FEEDER_WORKERS=8 python tools/train_surface_bench.py > $OUT/train_surface_8_workers.json 2>> $OUT/train_surface.err
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_depth.py::test_headline_shape_parity > $OUT/gpu_tests.log 2>&1
echo "gpu tests rc $?" >> $OUT/gpu_tests.log
ls -la $OUT
