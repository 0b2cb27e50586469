#!/bin/bash
# config 4's share with two library builds on the same box, + the K = 100 parity tests on the first one
echo "== K=100 tests"; timeout 900 python -m pytest tests -m gpu -q -x -k "k100 or config4 or shapes or edge or fuzz" 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2; do for lib in "$@"; do
  STM_LIB_PATH=$PWD/strutopy_amd/$lib timeout 600 python bench.py --config c4 --steps 8 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']), round(d['ms_per_step'],2), {k:round(v.get('avg_launch_ms'),2) for k,v in d['roofline']['kernels'].items()}, round(d['roofline']['estep_frac'],4))"
done; done
