bash tools/profile_round.sh r05 > gpurun_out/profile_round.log 2>&1
mkdir -p gpurun_out/fin
cp gpurun_out/r05/pmc_fetch_write_per_kernel.csv profiles/r05_pmc_fetch_write_per_kernel.csv
cp gpurun_out/r05/pmc_kernel_source_sha16.txt profiles/r05_pmc_kernel_source_sha16.txt
python bench.py --steps 10 --warmup 3 > gpurun_out/fin/bench_line.json 2> gpurun_out/fin/err.txt
python bench.py --steps 10 --warmup 3 --config3 --no-cpu-baseline > gpurun_out/fin/bench_line_config3.json 2>> gpurun_out/fin/err.txt
python bench.py --steps 10 --warmup 3 --config3 --config3-f32-loops --no-cpu-baseline > gpurun_out/fin/bench_line_config3_f32_loops.json 2>> gpurun_out/fin/err.txt
python bench.py --steps 10 --warmup 3 --force-allreduce --no-cpu-baseline > gpurun_out/fin/bench_line_one_rank_rccl.json 2>> gpurun_out/fin/err.txt
python bench.py --steps 10 --warmup 3 --force-allreduce --config3 --no-cpu-baseline > gpurun_out/fin/bench_line_one_rank_rccl_config3.json 2>> gpurun_out/fin/err.txt
