#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || true
rm -rf gpurun_out/tl; timeout 600 rocprofv3 --kernel-trace -d gpurun_out/tl -o tl --output-format csv -- python bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/tl_bench.json 2> gpurun_out/tl_bench.err
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 8
rm -rf gpurun_out/tl
