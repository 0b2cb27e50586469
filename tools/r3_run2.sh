#!/bin/bash
out=gpurun_out/${1:-r3b}; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^FAILED|passed|failed" $out/pytest.log | tail -20
python tools/r3_diag.py 2>&1 | tail -3
STM_POST_IMPL=1 python tools/r3_diag.py 2>&1 | tail -3
