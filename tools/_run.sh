cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_$c -o pmc -- python /root/repo/bench.py --steps 1 --warmup 1 --frames 48 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
cd /root/repo
python tools/pmc_summary.py gpurun_out/pmc_summary.csv gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
head -30 gpurun_out/pmc_summary.csv | cut -c1-160
