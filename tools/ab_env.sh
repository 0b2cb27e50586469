#!/bin/bash
# tools/ab_env.sh <VAR> <value> [<value> ...] [-- bench args]: the driver's bench with an environment switch at each value, twice, on one box
var=$1; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done; [ "$1" == "--" ] && shift
for i in 1 2; do for v in "${vals[@]}"; do
  env $var=$v timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --late-sample 0 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['value']), round(d['ms_per_step'],3), {k:round(v.get('avg_launch_ms'),3) for k,v in d['roofline']['kernels'].items()}, 'ELBO', d.get('elbo_trace', [None])[-1])"
done; done
