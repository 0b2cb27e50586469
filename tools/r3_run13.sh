#!/bin/bash
# DMA gather A/B: parity tests at K=50, then profile + bench
echo "== parity (K=50 paths)"; timeout 600 python -m pytest tests -m gpu -q -x -k "k50 or full_size or late or c2 or content or shapes or edge or stale" 2>&1 | tail -6
echo "== prof dma"; timeout 300 python tools/solver_prof.py 100000 10000 50 8 7 2>&1 | head -4
echo "== prof old"; STM_SOLVER_DMA=0 timeout 300 python tools/solver_prof.py 100000 10000 50 8 7 2>&1 | head -4
echo "== bench dma"; timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v.get('avg_launch_ms') for k,v in d['roofline']['kernels'].items()})"
echo "== bench old"; STM_SOLVER_DMA=0 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v.get('avg_launch_ms') for k,v in d['roofline']['kernels'].items()})"
