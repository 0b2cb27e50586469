#!/bin/bash
# A/B of two builds of the library on the driver's bench (12 steps) + the late-iteration parity tests
for lib in libstm_hip.so ${1:-libstm_hip_udlds.so}; do
  echo "== $lib"
  STM_LIB_PATH=$PWD/strutopy_amd/$lib timeout 200 python bench.py --steps 12 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v.get('avg_launch_ms') for k,v in d['roofline']['kernels'].items()}, d.get('late_check'))"
done
STM_LIB_PATH=$PWD/strutopy_amd/${1:-libstm_hip_udlds.so} timeout 300 python -m pytest tests -m gpu -q -x -k "late or full_size or k50 or reference or edge or shapes" 2>&1 | tail -3
