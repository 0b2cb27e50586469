#!/bin/bash
# usage: profile_r03.sh <tag>: what the driver times (bench.py --steps 20 --warmup 5), profiled:
#   1. rocprofv3 --kernel-trace --stats of that very command (+ per-EM-iteration split of the kernel times)
#   2. SQ counters (two --pmc passes) and HBM FETCH_SIZE / WRITE_SIZE (two more) over EM iterations 0-8, split 0 / 1-3 / 4 / 5+
tag=$1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag --output-format csv -- python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -1 gpurun_out/${tag}_bench.json | cut -c1-300
python tools/by_iteration.py trace gpurun_out/prof_$tag/*/${tag}_kernel_trace.csv 5 20 > gpurun_out/${tag}_steps20_by_iteration.txt 2>&1 || python tools/by_iteration.py trace gpurun_out/prof_$tag/${tag}_kernel_trace.csv 5 20 > gpurun_out/${tag}_steps20_by_iteration.txt 2>&1
cat gpurun_out/${tag}_steps20_by_iteration.txt
find gpurun_out/prof_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_steps20_kernel_stats.csv \;
B="python bench.py --steps 9 --warmup 0 --cpu-sample 0"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d gpurun_out/sq_${tag}_a -o p --output-format csv -- $B > gpurun_out/sq_${tag}_a.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM -d gpurun_out/sq_${tag}_b -o p --output-format csv -- $B > gpurun_out/sq_${tag}_b.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_${tag}_$c -o p --output-format csv -- $B > gpurun_out/pmc_${tag}_$c.log 2>&1
done
{ for p in sq_${tag}_a sq_${tag}_b pmc_${tag}_FETCH_SIZE pmc_${tag}_WRITE_SIZE; do
    f=$(find gpurun_out/$p -name "*counter_collection.csv" | head -1)
    echo "== $p ($f)"; python tools/by_iteration.py pmc $f 0 9
  done; } > gpurun_out/${tag}_counters_by_iteration.txt 2>&1
head -60 gpurun_out/${tag}_counters_by_iteration.txt
# one-iteration traffic file in the format bench.py reads (EM iteration 0 only: --steps 1)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc1_${tag}_$c -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/pmc1_${tag}_$c.log 2>&1
done
python tools/traffic_summary.py $(find gpurun_out/pmc1_${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc1_${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -1) gpurun_out/hbm_traffic_$tag.json | tail -12
# config 4's per-GPU share: kernel stats of `bench.py --config c4`
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_c4 -o ${tag}_c4 --output-format csv -- python bench.py --config c4 --steps 8 --warmup 2 > gpurun_out/${tag}_c4_bench.json 2> gpurun_out/${tag}_c4_bench.err
find gpurun_out/prof_${tag}_c4 -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_c4_kernel_stats.csv \;
python tools/by_iteration.py trace $(find gpurun_out/prof_${tag}_c4 -name "*kernel_trace.csv" | head -1) 2 8 > gpurun_out/${tag}_c4_by_iteration.txt 2>&1
cat gpurun_out/${tag}_c4_by_iteration.txt
# where the rest of an EM iteration goes: every dispatch of EM iteration 8 with its gaps
python tools/timeline.py $(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1) 12 > gpurun_out/${tag}_timeline_it7.txt 2>&1
cat gpurun_out/${tag}_timeline_it7.txt
