#!/bin/bash
timeout 100 python tools/r3_small.py 6000 2>&1 | tail -1
timeout 100 python tools/r3_small2.py 100000 2>&1 | tail -3
timeout 120 python tools/solver_prof.py 100000 10000 50 2 2>&1 | grep -E "^it1|post" | tail -4 | cut -c1-400
echo "chol subphases (update, load, panel, store):"; STM_POST_DEBUG=32 timeout 120 python tools/solver_prof.py 100000 10000 50 2 2>&1 | grep -E "post inverse" | tail -1
bash tools/r3_prof.sh 2>&1 | head -4 | cut -c1-150
