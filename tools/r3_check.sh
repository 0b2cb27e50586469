#!/bin/bash
# round-3 bring-up: parity tests, then A/B of the post kernels (new vs STM_POST_IMPL=1) on the bench workload
out=gpurun_out/${1:-r3a}
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 > $out/bench_new.json 2> $out/bench_new.err; tail -2 $out/bench_new.err
STM_POST_IMPL=1 timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 > $out/bench_v1.json 2> $out/bench_v1.err
python - <<PY
import json
for t in ("new","v1"):
    try:
        d=json.loads(open("$out/bench_%s.json"%t).read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], {k:v.get("avg_launch_ms") for k,v in d["roofline"].get("kernels",{}).items()} if isinstance(d["roofline"].get("kernels"),dict) else d["roofline"])
    except Exception as e: print(t,"failed",e)
PY
