#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06o
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
BENCH_ARGS=--no-surface bash tools/ab_one.sh persist_bwd "-DBSPLIT_C1=0" "-DBSPLIT_C1=1" > $OUT/ab_bsplit_c1.txt 2>&1
# parity of the split variant (left linked by the loop above is RECON=1; relink the plain split form first)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DBSPLIT_C1=1 -x hip -c multi_speaker_tts_amd/csrc/persist_bwd.hip -o multi_speaker_tts_amd/csrc/persist_bwd.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o multi_speaker_tts_amd/libmstts_hip.so multi_speaker_tts_amd/csrc/*.o
timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_depth.py -m gpu -q -k "bptt or test_depth_parity_train" > $OUT/parity_split.log 2>&1
tail -3 $OUT/parity_split.log; cat $OUT/ab_bsplit_c1.txt | cut -c1-330
