#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06d
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_persist.py tests/test_gpu_persist_infer.py tests/test_gpu_persist_lstm.py tests/test_gpu_speaker_trainer.py tests/test_gpu_taco1_trainer.py tests/test_gpu_thirdparty_pins.py tests/test_gpu_waveglow.py tests/test_gpu_model.py -m gpu -q -k "not test_train_step_parity" > $OUT/gpu_tests.log 2>&1
echo "gpu tests rc $?" >> $OUT/gpu_tests.log
BENCH_ARGS=--no-surface bash tools/ab_one.sh persist "-DPRE_EARLY=0" "-DPRE_EARLY=1" > $OUT/ab_pre_early.txt 2>&1
ls -la $OUT
