#!/bin/bash
# A/B of two compile-time variants ON THE SAME GPU BOX (box-to-box variance is ~1 %, more than most single changes):
#   bash tools/ab_bench.sh "-DVARIANT_A" "-DVARIANT_B" [reps]
# rebuilds the library with each flag in turn (all sources), runs bench.py after each, alternating `reps` times.
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "from multi_speaker_tts_amd import lib; lib.load()" > /dev/null 2>&1    # (a fresh snapshot may rebuild the library once on its first load: do that BEFORE the first variant is linked)
A=$1; B=$2; REPS=${3:-2}
build() {
  for f in multi_speaker_tts_amd/csrc/*.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $1 -x hip -c $f -o ${f%.hip}.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o multi_speaker_tts_amd/libmstts_hip.so multi_speaker_tts_amd/csrc/*.o
}
for rep in $(seq $REPS); do
  for v in "$A" "$B"; do
    build "$v"
    echo "$v: $(python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {k:round(x,2) for k,x in d['kernel_avg_us'].items()})")"
  done
done
