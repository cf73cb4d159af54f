#!/bin/bash
# A/B of compile-time variants of ONE source file on the same GPU box, by a kernel's average duration in a rocprofv3 trace of the bench:
#   bash tools/ab_kernel.sh lsa lsa_param_bwd_kernel "-DLSA_PARAM_WAVES=2" "-DLSA_PARAM_WAVES=3"
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "from multi_speaker_tts_amd import lib; lib.load()" > /dev/null 2>&1    # (a fresh snapshot may rebuild the library once on its first load: do that BEFORE the first variant is linked)
ROOT=$PWD
SRC=$1; KERN=$2; shift; shift
export TMPDIR=/tmp
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $v -x hip -c multi_speaker_tts_amd/csrc/$SRC.hip -o multi_speaker_tts_amd/csrc/$SRC.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o multi_speaker_tts_amd/libmstts_hip.so multi_speaker_tts_amd/csrc/*.o
  rm -rf /tmp/kt_ab
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_ab -o kt -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/kt_ab.log 2>&1)
  python tools/rocpd_stats.py /tmp/kt_ab/kt_results.db /tmp/kt_ab.csv
  echo "$v: $(grep -o '"ms_per_step": [0-9.]*' /tmp/kt_ab.log) $(python -c "
import csv,sys
for r in csv.reader(open('/tmp/kt_ab.csv')):
    if '$KERN' in r[0]: print(r[1], 'calls, avg', round(float(r[3])/1e3,1), 'us')")"
done
