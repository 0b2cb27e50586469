#!/bin/bash
# per-document cycle profile of two library builds on the same box: ab_prof.sh "<solver_prof args>" lib1 lib2
args=$1; shift
for lib in "$@"; do echo "== $lib"; STM_LIB_PATH=$PWD/strutopy_amd/$lib timeout 600 python tools/solver_prof.py $args 2>&1 | grep -E "^it|set-up|post kernel cycles" | cut -c1-330; done
