#!/bin/bash
# usage: pmc_run.sh <tag> <counters...>   -- one rocprofv3 --pmc pass over a 20k-document bench step
tag=$1; shift
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_$tag -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --docs 20000 --cpu-sample 0 > gpurun_out/pmc_$tag.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_$tag/p_counter_collection.csv
