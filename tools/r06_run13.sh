#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06p
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
A="-DBPTT_QOWN=1"; B="-DBPTT_QOWN=1 -DBPTT_W0LDS=1"; C="-DBPTT_QOWN=1 -DBPTT_W0LDS=1 -DBSPLIT_C1=1"; D="-DBPTT_QOWN=1 -DBPTT_W0LDS=1 -DBSPLIT_C1=1 -DBSPLIT_C0=1"
BENCH_ARGS=--no-surface bash tools/ab_one.sh persist_bwd "-DBPTT_QOWN=0" "$A" "$B" "$C" "$D" > $OUT/ab_qown.txt 2>&1
# parity on the last-built variant (D), then on C
timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_depth.py -m gpu -q -x -k "bptt or test_depth_parity_train" > $OUT/parity_D.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $C -x hip -c multi_speaker_tts_amd/csrc/persist_bwd.hip -o multi_speaker_tts_amd/csrc/persist_bwd.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o multi_speaker_tts_amd/libmstts_hip.so multi_speaker_tts_amd/csrc/*.o
timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_depth.py -m gpu -q -x -k "bptt or test_depth_parity_train" > $OUT/parity_C.log 2>&1
tail -3 $OUT/parity_D.log; tail -3 $OUT/parity_C.log; cat $OUT/ab_qown.txt | cut -c1-300
