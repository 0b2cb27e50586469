#!/bin/bash
# usage: collect_profiles.sh <tag>: copies the summaries tools/profile_<tag>.sh left under gpurun_out/ into profiles/ (tracked) and folds the
# per-workload traffic / compute-counter entries into profiles/hbm_traffic.json / profiles/compute_counters.json
tag=$1
cd "$(dirname "$0")/.." || exit 1
for f in steps20_by_iteration.txt steps20_kernel_stats.csv counters_by_iteration.txt c2_counters_by_iteration.txt c4_counters_by_iteration.txt c5_counters_by_iteration.txt \
         c4_by_iteration.txt c4_kernel_stats.csv timeline_it7.txt c4_corpus_shards.json solver_prof_it7.txt solver_prof_it12.txt solver_prof_c4.txt; do
  [ -f gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${tag}_$f
done
for f in steps20 steps50 c4 c5 docs12500 rccl1 rccl1_single rccl1_docs12500 rccl1_single_docs12500 c4_1M; do
  [ -f gpurun_out/${tag}_${f}_bench.json ] && grep '^{"metric' gpurun_out/${tag}_${f}_bench.json | tail -1 > profiles/${tag}_${f}_bench.json
done
ls gpurun_out/hbm_traffic_${tag}_*.json > /dev/null 2>&1 && python tools/traffic_merge.py profiles/hbm_traffic.json gpurun_out/hbm_traffic_${tag}_*.json
ls gpurun_out/compute_${tag}_*.json > /dev/null 2>&1 && python tools/traffic_merge.py profiles/compute_counters.json gpurun_out/compute_${tag}_*.json
ls -la profiles/${tag}_* | head -40
