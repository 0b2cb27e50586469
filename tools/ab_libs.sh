#!/bin/bash
# A/B of library builds: per-document solver cycles at EM iteration 7 and the driver's bench
for lib in "$@"; do
  echo "== $lib"
  STM_LIB_PATH=$PWD/strutopy_amd/$lib timeout 300 python tools/solver_prof.py 100000 10000 50 8 7 2>&1 | head -3 | cut -c1-330
  STM_LIB_PATH=$PWD/strutopy_amd/$lib timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v.get('avg_launch_ms') for k,v in d['roofline']['kernels'].items()})"
done
