#!/bin/bash
out=gpurun_out/${1:-r3c}; mkdir -p $out
timeout 300 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^FAILED|passed|failed|rc=" $out/pytest.log | tail -5
timeout 120 python tools/solver_prof.py 100000 10000 50 2 2>&1 | grep -E "^it1|post" | tail -4 | cut -c1-400
for w in 8 11; do echo "WG/CU $w"; STM_POST_MAX_WG_PER_CU=$w timeout 120 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --late-sample 0 2>$out/bench_err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], {k:v.get('avg_launch_ms') for k,v in d['roofline']['kernels'].items()})"; done
tail -3 $out/bench_err.txt
cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r3 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --late-sample 0 > /dev/null 2>&1; find /tmp/prof -name "*kernel_stats*" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-160'
