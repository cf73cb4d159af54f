#!/bin/bash
# Everything the round's profiles/ directory is built from, run on the GPU box:  bash tools/profile_round.sh r02
#   1. bench line (fp32 headline) and the config-3 line
#   2. rocprofv3 --kernel-trace of the same bench command -> per-kernel stats CSV
#   3. PMC passes, each in its own run with --kernel-trace only: FETCH_SIZE, WRITE_SIZE (HBM-side traffic per kernel), then
#      SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE (matrix-core busy) - all at the FULL config-2 size
#   4. Audio.melspectrogram timing + its kernel rows
#   5. free-running decoder (tools/infer_bench.py) timing + its kernel rows
#   6. device idle time inside a step, every GEMM call of a step replayed on its own, the BiLSTM recurrence bench, the auxiliary trainers,
#      per-workgroup stage stamps of the persistent decoder launches
# Outputs land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --steps 10 --warmup 3 > $OUT/bench_line.json 2> $OUT/bench_line.err
python $ROOT/bench.py --steps 10 --warmup 3 --config3 --no-cpu-baseline > $OUT/bench_line_config3.json 2>> $OUT/bench_line.err
python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --force-allreduce 2>> $OUT/bench_line.err | tail -1 > $OUT/bench_line_one_rank_rccl.json
python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --force-allreduce --config3 2>> $OUT/bench_line.err | tail -1 > $OUT/bench_line_one_rank_rccl_config3.json
python $ROOT/tools/train_surface_bench.py > $OUT/train_surface.json 2>> $OUT/bench_line.err
MSTTS_FEEDER_WORKERS=0 python $ROOT/tools/train_surface_bench.py > $OUT/train_surface_loader_in_the_producer_thread.json 2>> $OUT/bench_line.err
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/kt.log 2>&1
python $ROOT/tools/rocpd_stats.py $OUT/kt/kt_results.db $OUT/train_step_kernel_stats.csv
rocprofv3 --kernel-trace -d $OUT/kt3 -o kt -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --config3 > $OUT/kt3.log 2>&1
python $ROOT/tools/rocpd_stats.py $OUT/kt3/kt_results.db $OUT/train_step_kernel_stats_config3.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_$c.log 2>&1
done
python $ROOT/tools/pmc_summary.py $OUT/pmc_fetch_write_per_kernel.csv $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
(cd $ROOT && python -c "import bench; print(bench.persist_source_sha16())") > $OUT/pmc_kernel_source_sha16.txt     # what the pass measured: bench.py reports traffic_stale against it
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmcm_$c -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmcm_$c.log 2>&1
done
python $ROOT/tools/pmc_summary.py $OUT/pmc_mfma_busy_per_kernel.csv $OUT/pmcm_SQ_VALU_MFMA_BUSY_CYCLES $OUT/pmcm_SQ_BUSY_CYCLES $OUT/pmcm_GRBM_GUI_ACTIVE
python $ROOT/tools/stft_bench.py > $OUT/stft_mel_line.json 2> $OUT/stft.err
rocprofv3 --kernel-trace -d $OUT/kt_stft -o kt -- python $ROOT/tools/stft_bench.py > $OUT/kt_stft.log 2>&1
python $ROOT/tools/rocpd_stats.py $OUT/kt_stft/kt_results.db $OUT/stft_mel_kernel_stats.csv
python $ROOT/tools/infer_bench.py > $OUT/infer_line.txt 2> $OUT/infer.err
rocprofv3 --kernel-trace -d $OUT/kt_inf -o kt -- python $ROOT/tools/infer_bench.py > $OUT/kt_inf.log 2>&1
python $ROOT/tools/rocpd_stats.py $OUT/kt_inf/kt_results.db $OUT/infer_kernel_stats.csv
python $ROOT/tools/rocpd_gaps.py $OUT/kt/kt_results.db $OUT/train_step_gaps.txt
python $ROOT/tools/rocpd_timeline.py $OUT/kt/kt_results.db $OUT/train_step_timeline.txt
rocprofv3 --kernel-trace -d $OUT/ktr -o kt -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --force-allreduce > $OUT/ktr.log 2>&1
python $ROOT/tools/rocpd_timeline.py $OUT/ktr/kt_results.db $OUT/one_rank_rccl_timeline.txt
python $ROOT/tools/gemm_step_profile.py > $OUT/gemm_step_profile.txt 2> /dev/null
python $ROOT/tools/lstm_bench.py > $OUT/lstm_bench.txt 2> /dev/null
python $ROOT/tools/aux_trainers_bench.py > $OUT/aux_trainers.txt 2> /dev/null
python $ROOT/tools/persist_stamps.py > $OUT/persist_stamps_per_workgroup.txt 2> /dev/null
rm -rf $OUT/ktr $OUT/kt $OUT/kt3 $OUT/kt_stft $OUT/kt_inf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmcm_*
ls -la $OUT
