#!/bin/bash
# usage: profile_r06.sh <tag> [quick]: the evidence behind bench.py's numbers, one workload at a time
#   1. what the driver times (bench.py --steps 20 --warmup 5) under rocprofv3 --kernel-trace --stats, split per EM iteration
#   2. SQ counters (two --pmc passes) over EM iterations 0-8 of that workload
#   3. HBM traffic per EM iteration (FETCH_SIZE / WRITE_SIZE, separate passes) for c2 (20 its), c4's share (8) and c5 (20)
#   4. config 4's share: kernel trace + SQ counters (post_big2_kernel, the DIRECT solver)
#   5. every dispatch of one late EM iteration with its gaps
#   6. compute-side counter entries per workload (tools/compute_collect.py) and the bench lines of c5, --docs 12500, --allreduce rccl with both
#      exchanges (split | single), config 4 as one corpus; the per-document cycle profile (tools/solver_prof.py: the -DSTM_TESTING build)
tag=$1; quick=$2
export TMPDIR=/tmp
mkdir -p gpurun_out
S2=20; S4=8; S5=20
if [ -n "$quick" ]; then S2=3; S4=3; S5=3; fi
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag --output-format csv -- python bench.py --steps $S2 --warmup 5 > gpurun_out/${tag}_steps20_bench.json 2> gpurun_out/${tag}_bench.err
grep '^{"metric' gpurun_out/${tag}_steps20_bench.json | tail -1 | cut -c1-300
python tools/by_iteration.py trace $(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1) 5 $S2 > gpurun_out/${tag}_steps20_by_iteration.txt 2>&1
cat gpurun_out/${tag}_steps20_by_iteration.txt
find gpurun_out/prof_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_steps20_kernel_stats.csv \;
python tools/timeline.py $(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1) 12 > gpurun_out/${tag}_timeline_it7.txt 2>&1
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
SQB="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM"
counters() {   # <name> <bench args> <iterations>
  local name=$1 bargs=$2 its=$3
  timeout 900 rocprofv3 --kernel-trace --pmc $SQA -d gpurun_out/sq_${tag}_${name}_a -o p --output-format csv -- python bench.py $bargs --steps $its --warmup 0 --cpu-sample 0 --late-sample 0 > gpurun_out/sq_${tag}_${name}_a.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc $SQB -d gpurun_out/sq_${tag}_${name}_b -o p --output-format csv -- python bench.py $bargs --steps $its --warmup 0 --cpu-sample 0 --late-sample 0 > gpurun_out/sq_${tag}_${name}_b.log 2>&1
  python tools/compute_collect.py gpurun_out/compute_${tag}_${name}.json $(find gpurun_out/sq_${tag}_${name}_a -name "*counter_collection.csv" | head -1) \
      $(find gpurun_out/sq_${tag}_${name}_b -name "*counter_collection.csv" | head -1) $4 $5 $6 $7 $8 0 $its \
      "rocprofv3 --kernel-trace --pmc SQ passes of bench.py $bargs --steps $its --warmup 0 --cpu-sample 0 --late-sample 0 (tools/profile_r06.sh $tag)" > gpurun_out/compute_${tag}_${name}.txt 2>&1
  { for p in sq_${tag}_${name}_a sq_${tag}_${name}_b; do
      f=$(find gpurun_out/$p -name "*counter_collection.csv" | head -1)
      echo "== $p ($f): bench.py $bargs --steps $its --warmup 0"; python tools/by_iteration.py pmc $f 0 $its
    done; } > gpurun_out/${tag}_${name}_counters_by_iteration.txt 2>&1
}
traffic() {    # <name> <bench args> <iterations> docs vocab topics words levels
  local name=$1 bargs=$2 its=$3
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_${tag}_${name}_$c -o p --output-format csv -- python bench.py $bargs --steps $its --warmup 0 --cpu-sample 0 --late-sample 0 > gpurun_out/pmc_${tag}_${name}_$c.log 2>&1
  done
  python tools/traffic_collect.py gpurun_out/hbm_traffic_${tag}_${name}.json $(find gpurun_out/pmc_${tag}_${name}_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
      $(find gpurun_out/pmc_${tag}_${name}_WRITE_SIZE -name "*counter_collection.csv" | head -1) $4 $5 $6 $7 $8 \
      "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py $bargs --steps $its --warmup 0 --cpu-sample 0 --late-sample 0 (tools/profile_r06.sh $tag)"
}
counters c2 "" 20 100000 10000 50 150 1
traffic c2 "" $S2 100000 10000 50 150 1
traffic c4 "--config c4" $S4 125000 50000 100 150 1
traffic c5 "--config c5" $S5 100000 10000 50 150 2
# config 4's per-GPU share: kernel stats + counters
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_c4 -o ${tag}_c4 --output-format csv -- python bench.py --config c4 --steps $S4 --warmup 2 > gpurun_out/${tag}_c4_bench.json 2> gpurun_out/${tag}_c4_bench.err
find gpurun_out/prof_${tag}_c4 -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_c4_kernel_stats.csv \;
python tools/by_iteration.py trace $(find gpurun_out/prof_${tag}_c4 -name "*kernel_trace.csv" | head -1) 2 $S4 > gpurun_out/${tag}_c4_by_iteration.txt 2>&1
cat gpurun_out/${tag}_c4_by_iteration.txt
counters c4 "--config c4" $S4 125000 50000 100 150 1
counters c5 "--config c5" $S5 100000 10000 50 150 2
# bench lines kept for the figures DESIGN quotes: config 5, config 2's 8-GPU share, the with-communicator iteration, config 4 as ONE corpus
timeout 600 python bench.py --config c5 --steps 20 --warmup 5 > gpurun_out/${tag}_c5_bench.json 2> gpurun_out/${tag}_c5_bench.err
timeout 600 python bench.py --docs 12500 --steps 20 --warmup 5 > gpurun_out/${tag}_docs12500_bench.json 2> gpurun_out/${tag}_docs12500_bench.err
timeout 600 python bench.py --gpus 1 --allreduce rccl --steps 20 --warmup 5 > gpurun_out/${tag}_rccl1_bench.json 2> gpurun_out/${tag}_rccl1_bench.err
timeout 600 python bench.py --gpus 1 --allreduce rccl --exchange single --steps 20 --warmup 5 > gpurun_out/${tag}_rccl1_single_bench.json 2> gpurun_out/${tag}_rccl1_single_bench.err
timeout 600 python bench.py --docs 12500 --gpus 1 --allreduce rccl --steps 20 --warmup 5 > gpurun_out/${tag}_rccl1_docs12500_bench.json 2> gpurun_out/${tag}_rccl1_docs12500_bench.err
timeout 600 python bench.py --docs 12500 --gpus 1 --allreduce rccl --exchange single --steps 20 --warmup 5 > gpurun_out/${tag}_rccl1_single_docs12500_bench.json 2> gpurun_out/${tag}_rccl1_single_docs12500_bench.err
timeout 600 python bench.py --steps 50 --warmup 5 --cpu-sample 0 > gpurun_out/${tag}_steps50_bench.json 2> gpurun_out/${tag}_steps50_bench.err
timeout 300 python tools/solver_prof.py 100000 10000 50 8 7 > gpurun_out/${tag}_solver_prof_it7.txt 2>&1
timeout 600 python tools/solver_prof.py 125000 50000 100 6 5 > gpurun_out/${tag}_solver_prof_c4.txt 2>&1
timeout 900 python bench.py --config c4 --docs 1000000 --steps 8 --warmup 2 --cpu-sample 2000 > gpurun_out/${tag}_c4_1M_bench.json 2> gpurun_out/${tag}_c4_1M_bench.err
timeout 900 python tools/c4_corpus_shards.py > gpurun_out/${tag}_c4_corpus_shards.json 2> gpurun_out/${tag}_c4_corpus_shards.err
head -40 gpurun_out/${tag}_c4_counters_by_iteration.txt
ls gpurun_out | grep $tag | head -40
