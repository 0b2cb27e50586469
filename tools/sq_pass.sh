#!/bin/bash
# usage: sq_pass.sh <tag>: SQ counters of the default workload (100k documents, one EM iteration) in two --pmc passes:
# wave / wait / busy cycles, then the fp64 MFMA and LDS counters.  Summaries via tools/pmc_summary.py.
tag=$1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d gpurun_out/sq_${tag}_a -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/sq_${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES -d gpurun_out/sq_${tag}_b -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/sq_${tag}_b.log 2>&1
for p in a b; do python tools/pmc_summary.py gpurun_out/sq_${tag}_$p/p_counter_collection.csv; done
