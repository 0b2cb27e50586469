#!/bin/bash
# usage: profile_round.sh <tag>: kernel-trace + stats of the default bench, then FETCH_SIZE / WRITE_SIZE passes
tag=$1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag --output-format csv -- python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
tail -1 gpurun_out/bench_$tag.json | cut -c1-400
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_${tag}_$c -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/pmc_${tag}_$c.log 2>&1
done
python tools/traffic_summary.py gpurun_out/pmc_${tag}_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_${tag}_WRITE_SIZE/p_counter_collection.csv gpurun_out/hbm_traffic_$tag.json
ls gpurun_out/prof_$tag
