#!/bin/bash
# usage: profile_c4.sh <tag>: BASELINE config 4's per-GPU share (125k documents, V = 50k, K = 100) under rocprofv3 --kernel-trace --stats
tag=$1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_c4 -o ${tag}_c4 --output-format csv -- python bench.py --config c4 --steps 8 --warmup 2 > gpurun_out/${tag}_c4_bench.json 2> gpurun_out/${tag}_c4_bench.err
tail -1 gpurun_out/${tag}_c4_bench.json | cut -c1-400
find gpurun_out/prof_${tag}_c4 -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_c4_kernel_stats.csv \;
python tools/by_iteration.py trace $(find gpurun_out/prof_${tag}_c4 -name "*kernel_trace.csv" | head -1) 2 8 > gpurun_out/${tag}_c4_by_iteration.txt 2>&1
cat gpurun_out/${tag}_c4_by_iteration.txt
head -8 gpurun_out/${tag}_c4_kernel_stats.csv | cut -c1-200
