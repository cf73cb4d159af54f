cd /root/repo
timeout 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value']); print(d['kernel_avg_us']); print(d['roofline']['frac'])"
