#!/bin/bash
# round 6, first GPU call: the new tests, the baseline bench lines of this box, and where the fills of a step are
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06a
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "reference_widths or one_flow_at_reference_width or two_deferred or depth_parity_train_bf16 or test_depth_parity_train or gemm or two_ranks or one_rank or dist" > $OUT/new_tests.log 2>&1
echo "new tests rc $?" >> $OUT/new_tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_line.json 2> $OUT/bench_line.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --force-allreduce > $OUT/bench_line_one_rank_rccl.json 2>> $OUT/bench_line.err
MSTTS_ASYNC_AGREE=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --force-allreduce > $OUT/bench_line_one_rank_rccl_blocking_agree.json 2>> $OUT/bench_line.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --config3 > $OUT/bench_line_config3.json 2>> $OUT/bench_line.err
cd /tmp
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/kt.log 2>&1
python $ROOT/tools/rocpd_timeline.py $OUT/kt/kt_results.db $OUT/step_timeline.txt
python $ROOT/tools/rocpd_stats.py $OUT/kt/kt_results.db $OUT/train_step_kernel_stats.csv
rocprofv3 --kernel-trace -d $OUT/ktr -o kt -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --force-allreduce > $OUT/ktr.log 2>&1
python $ROOT/tools/rocpd_timeline.py $OUT/ktr/kt_results.db $OUT/step_timeline_one_rank_rccl.txt
rm -rf $OUT/kt $OUT/ktr
ls -la $OUT
