#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06t
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
C="-DBPTT_QOWN=1 -DBPTT_W0LDS=1 -DBSPLIT_C1=1"
CE="-DBPTT_QOWN=1 -DBPTT_W0LDS=1 -DBSPLIT_C1=1 -DBPTT_DQ_EARLY=1"
D2E="-DBPTT_QOWN=1 -DBPTT_W0LDS=1 -DBSPLIT_C1=1 -DBSPLIT_C0=1 -DBPTT_KEYS_PER_STEP=1 -DBPTT_LATE_OP0=1 -DBPTT_DQ_EARLY=1"
BENCH_ARGS=--no-surface bash tools/ab_one.sh persist_bwd "-DBPTT_QOWN=0" "$C" "$CE" "$D2E" > $OUT/ab.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $CE -x hip -c multi_speaker_tts_amd/csrc/persist_bwd.hip -o multi_speaker_tts_amd/csrc/persist_bwd.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o multi_speaker_tts_amd/libmstts_hip.so multi_speaker_tts_amd/csrc/*.o
timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_depth.py -m gpu -q -k "bptt or test_depth_parity_train" > $OUT/parity_CE.log 2>&1
tail -3 $OUT/parity_CE.log | cut -c1-200; grep -v "roofline" $OUT/ab.txt | cut -c1-250
