#!/bin/bash
# dev: timing ablations of the split GEMM (wrong results by construction): which of split arithmetic / LDS writes / LDS reads the tile time follows.
# Run on the GPU box: bash tools/gemm_ablate.sh > gpurun_out/gemm_ablate.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for v in "" "-DGS_ABL_NOSPLIT" "-DGS_ABL_NOSPLIT -DGS_ABL_WRITE1" "-DGS_ABL_READ1" "-DGS_ABL_NOSPLIT -DGS_ABL_WRITE1 -DGS_ABL_READ1"; do
  export MSTTS_EXTRA_HIPCC_FLAGS="$v"
  python -c "from multi_speaker_tts_amd import build; build.build()" > /dev/null 2>&1
  echo "=== flags: [$v]"
  python tools/gemm_ab.py --quick 2>/dev/null | sed 's/| f32 mfma.*//'
done
