#!/bin/bash
# development loop for the persistent decoder kernels on a GPU box: their own tests first (under a time limit), then the
# reference-width parity cases, then a bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_persist.py -x -q -m gpu ${PERSIST_K:+-k "$PERSIST_K"} 2>&1 | tail -40 > gpurun_out/persist_tests.log
cat gpurun_out/persist_tests.log
if [ -z "$SKIP_PARITY" ]; then
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "train_step_parity and not bf16" 2>&1 | tail -15 > gpurun_out/persist_parity.log
cat gpurun_out/persist_parity.log
fi
[ -n "$SKIP_BENCH" ] || timeout 600 python bench.py --steps 10 --warmup 3 ${BENCH_ARGS} 2>&1 | tail -5 > gpurun_out/persist_bench.log
cat gpurun_out/persist_bench.log
