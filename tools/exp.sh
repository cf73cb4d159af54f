cd $GRAFT_REPO_ROOT
export MSTTS_EXTRA_HIPCC_FLAGS="$1"
python -m multi_speaker_tts_amd.build > /dev/null 2>&1
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/exp.json 2> gpurun_out/exp.err
tail -3 gpurun_out/exp.err
