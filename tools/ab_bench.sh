#!/bin/bash
# same-box A/B of library builds on the driver's bench: ab_bench.sh [-t "pytest -k expression"] lib1 lib2 ... (three rounds)
if [ "$1" = "-t" ]; then echo "== tests ($2)"; timeout 900 python -m pytest tests -m gpu -q -x -k "$2" 2>&1 | grep -E "passed|failed" | tail -2; shift 2; fi
for i in 1 2 3; do for lib in "$@"; do
  STM_LIB_PATH=$PWD/strutopy_amd/$lib timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']), round(d['ms_per_step'],3), {k:round(v.get('avg_launch_ms'),3) for k,v in d['roofline']['kernels'].items()})"
done; done
