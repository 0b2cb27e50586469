#!/bin/bash
timeout 100 python tools/r3_small.py 6000 2>&1 | tail -1
timeout 100 python tools/r3_small2.py 100000 2>&1 | tail -2
bash tools/r3_prof.sh 2>&1 | grep -E "beta_ss|post_kernel" | cut -c1-140
