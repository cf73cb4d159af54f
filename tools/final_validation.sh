#!/bin/bash
# final pass of a round on HEAD: whole GPU suite, smoke, the profile round (usage: gpurun --timeout 3600 -- bash tools/final_validation.sh [round tag, default r06])
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
OUT=$ROOT/gpurun_out/${TAG}final
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1
echo "gpu tests rc $?" >> $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
tail -4 $OUT/gpu_tests.log | cut -c1-200; tail -1 $OUT/smoke.log
