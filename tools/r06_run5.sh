#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06e
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_waveglow.py tests/test_gpu_model.py tests/test_gpu_depth.py tests/test_gpu_persist_lstm.py -m gpu -q -s -k "test_persistent_is_deterministic or one_flow_at_reference_width or reference_widths or depth_parity_train_bf16 or two_deferred or variable_length" > $OUT/gpu_tests.log 2>&1
echo "gpu tests rc $?" >> $OUT/gpu_tests.log
bash tools/profile_round.sh r06 > $OUT/profile_round.log 2>&1
ls -la $OUT
