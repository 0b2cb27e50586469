#!/bin/bash
# usage: collect_profiles.sh <tag>: copies the summaries tools/profile_<tag>.sh left under gpurun_out/ into profiles/
tag=$1
for f in steps20_by_iteration.txt steps20_kernel_stats.csv counters_by_iteration.txt c4_by_iteration.txt c4_kernel_stats.csv timeline_it7.txt; do
  [ -f gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${tag}_$f
done
tail -1 gpurun_out/${tag}_bench.json > profiles/${tag}_steps20_bench.json
tail -1 gpurun_out/${tag}_c4_bench.json > profiles/${tag}_c4_bench.json
[ -f gpurun_out/hbm_traffic_$tag.json ] && cp gpurun_out/hbm_traffic_$tag.json profiles/hbm_traffic.json
ls -la profiles/${tag}_* profiles/hbm_traffic.json
