#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python -c "from multi_speaker_tts_amd import lib; lib.load()" > /dev/null 2>&1
A="-DBPTT_QOWN=1 -DBPTT_W0LDS=1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $A -x hip -c multi_speaker_tts_amd/csrc/persist_bwd.hip -o multi_speaker_tts_amd/csrc/persist_bwd.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o multi_speaker_tts_amd/libmstts_hip.so multi_speaker_tts_amd/csrc/*.o
MSTTS_ARENA_POISON=1 python tools/r06_dbg.py 2>&1 | tail -14
