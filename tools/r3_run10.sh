#!/bin/bash
# duplicate-device RCCL failure first (may hang: tight timeout), then new tests, fixed cost, full suite
mkdir -p gpurun_out
echo "== rccl duplicate"; timeout 240 python -m pytest tests/test_gpu_round2.py -m gpu -q -x -k "explicit_rccl" 2>&1 | tail -15
echo "== new"; timeout 400 python -m pytest tests/test_gpu_round2.py -m gpu -q -x -k "three_hundred or bench_launches or two_ranks or one_rank" 2>&1 | tail -15
echo "== small"; timeout 200 python bench.py --docs 12500 --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1
echo "== default"; timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1
echo "== suite"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
