#!/bin/bash
# kernel stats of the default bench (3 steps): per-kernel average durations
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r3 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps ${1:-3} --warmup 1 --cpu-sample 0 --late-sample 0 > /tmp/prof_bench.json 2>/tmp/prof_err.txt
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && { find /tmp/prof -type f | head; tail -5 /tmp/prof_err.txt; }
[ -n "$f" ] && head -14 "$f" | cut -c1-170
