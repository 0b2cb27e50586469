#!/bin/bash
for r in 256 1024 4096; do for f in 0 1; do echo "replicas $r ablate $f"; STM_SIGMA_REPLICAS=$r STM_POST_DEBUG=$f python tools/solver_prof.py 100000 10000 50 2 2>&1 | grep -E "^it1|post" | tail -3 | cut -c1-330; done; done
