#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06q
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
python -c "from multi_speaker_tts_amd import lib; lib.load()" > /dev/null 2>&1
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_persist.py -m gpu -q -k "bptt_equals" 2>&1 | tail -4 | cut -c1-200 >> $OUT/base.txt; done
C="-DBPTT_QOWN=1 -DBPTT_W0LDS=1 -DBSPLIT_C1=1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $C -x hip -c multi_speaker_tts_amd/csrc/persist_bwd.hip -o multi_speaker_tts_amd/csrc/persist_bwd.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o multi_speaker_tts_amd/libmstts_hip.so multi_speaker_tts_amd/csrc/*.o
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_persist.py -m gpu -q -k "bptt_equals" 2>&1 | grep -i "failed\|passed\|AssertionError: {" | cut -c1-300 >> $OUT/varC.txt; done
A="-DBPTT_QOWN=1 -DBPTT_W0LDS=1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $A -x hip -c multi_speaker_tts_amd/csrc/persist_bwd.hip -o multi_speaker_tts_amd/csrc/persist_bwd.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o multi_speaker_tts_amd/libmstts_hip.so multi_speaker_tts_amd/csrc/*.o
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_persist.py -m gpu -q -k "bptt_equals" 2>&1 | grep -i "failed\|passed\|AssertionError: {" | cut -c1-300 >> $OUT/varB.txt; done
echo BASE; cat $OUT/base.txt; echo VARC; cat $OUT/varC.txt; echo VARB; cat $OUT/varB.txt
