#!/bin/bash
# alternating A/B of one environment switch on the headline step: tools/ab_env.sh VAR A B [rounds] [extra bench flags]
VAR=$1; A=$2; B=$3; R=${4:-2}; shift 4
for i in $(seq $R); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 30 --warmup 5 --no-surface --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v ms_per_step %.3f' % d['ms_per_step'])"
  done
done
