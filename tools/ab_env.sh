#!/bin/bash
# A/B of one environment switch on the same box: ab_env.sh VAR valueA valueB [repeats]
v=$1; a=$2; b=$3; n=${4:-3}
for i in $(seq $n); do
  for x in $a $b; do
    env $v=$x timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v=$x', round(d['value']), round(d['ms_per_step'],3), {k:round(v.get('avg_launch_ms'),3) for k,v in d['roofline']['kernels'].items()})"
  done
done
