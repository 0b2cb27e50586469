#!/bin/bash
# tools/ab_w4.sh <lib>: post_big2_kernel with two / four waves per document (STM_POST_BIG2_WAVES) on one box -- the K > 64 parity tests with
# four waves, then config 4's share with both, twice
export STM_LIB_PATH=$PWD/strutopy_amd/$1
echo "== K > 64 tests, four waves"; STM_POST_BIG2_WAVES=4 timeout 900 python -m pytest tests -m gpu -q -x -k "k100 or k70 or above_64 or config4 or shapes or edge or fuzz" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for i in 1 2; do for w in 2 4; do
  STM_POST_BIG2_WAVES=$w timeout 600 python bench.py --config c4 --steps 8 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves $w', round(d['value']), round(d['ms_per_step'],2), {k:round(v.get('avg_launch_ms'),2) for k,v in d['roofline']['kernels'].items()}, round(d['roofline']['estep_frac'],4), 'ELBO', d.get('elbo_trace', [None])[-1])"
done; done
