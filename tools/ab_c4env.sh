#!/bin/bash
# tools/ab_c4env.sh <VAR> <value> [<value> ...]: config 4's share with an environment switch at each value, twice, on one box; K > 64 tests first
var=$1; shift
echo "== K > 64 tests"; timeout 900 python -m pytest tests -m gpu -q -x -k "k100 or k70 or above_64 or config4 or shapes or edge or fuzz or general_post" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for i in 1 2; do for v in "$@"; do
  env $var=$v timeout 600 python bench.py --config c4 --steps 8 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['value']), round(d['ms_per_step'],2), {k:round(v.get('avg_launch_ms'),2) for k,v in d['roofline']['kernels'].items()}, round(d['roofline']['estep_frac'],4), 'ELBO', d.get('elbo_trace', [None])[-1])"
done; done
