#!/bin/bash
# A/B of compile-time variants of ONE source file on the same GPU box:
#   bash tools/ab_one.sh persist_bwd "-DEARLY_DM1=31" "-DEARLY_DM1=16" ...   (each variant is built and benched twice, alternating; BENCH_ARGS=--config3 for the bf16 instantiations)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "from multi_speaker_tts_amd import lib; lib.load()" > /dev/null 2>&1    # (a fresh snapshot may rebuild the library once on its first load: do that BEFORE the first variant is linked)
SRC=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $v -x hip -c multi_speaker_tts_amd/csrc/$SRC.hip -o multi_speaker_tts_amd/csrc/$SRC.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o multi_speaker_tts_amd/libmstts_hip.so multi_speaker_tts_amd/csrc/*.o
    echo "$v: $(python bench.py --steps 10 --warmup 3 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],2), {k:round(x,2) for k,x in d['kernel_avg_us'].items()})
for k in ('roofline','bptt_persistent'):
    s=d.get(k) or {}
    print('   ',k, round(s.get('frame_us') or 0,2), [round(b,2) for a,b in (s.get('stage_us') or {}).items()])
")"
  done
done
