run() { python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), [ (round(t['solver_kernel_ms'],1), round(t['post_kernel_ms'],1)) for t in d['per_step']], d['elbo_trace'])"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
echo mode0-two-wave; run
echo mode3-one-wave; STM_SOLVER_MODE=3 run
echo "mode3 tests"; STM_SOLVER_MODE=3 timeout 900 python -m pytest tests -m gpu -x -q -k "estep_matches or shapes" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
