#!/bin/bash
# Same-box A/B of HIP / ROCr runtime settings that could change what a dependent launch costs (run on the GPU box: bash tools/env_ab.sh).
# Result of the round (ms per train step, two passes): default 87.6 / 88.0, HIP_FORCE_DEV_KERNARG=0 100.5 / 100.8 (kernel-argument
# blocks in host memory: +2 us on every one of the 6 408 launches), =1 88.0 / 88.1 (the default), HSA_ENABLE_INTERRUPT=0 88.1 / 88.1,
# ROC_ACTIVE_WAIT_TIMEOUT=1000 88.1 / 88.0.
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in "X=1" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=1000"; do
  echo "== $v"; env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done; done
