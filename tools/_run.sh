cd /root/repo
timeout 600 python -m pytest tests/test_gpu_waveglow.py -m gpu -q -x 2>&1 | tail -4
timeout 300 python tools/waveglow_bench.py 2>&1 | tail -1
N=8 timeout 300 python tools/waveglow_bench.py 2>&1 | tail -1
