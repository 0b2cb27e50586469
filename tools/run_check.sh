#!/bin/bash
# the round's usual confirmation on the GPU box: the -m gpu suite, the timeline of one EM iteration, C3's per-GPU share and the driver's bench
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
bash tools/trace_timeline.sh
echo "== small"; timeout 200 python bench.py --docs 12500 --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v.get('avg_launch_ms') for k,v in d['roofline']['kernels'].items()})"
echo "== default"; timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v.get('avg_launch_ms') for k,v in d['roofline']['kernels'].items()})"
