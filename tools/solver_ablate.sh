run() { python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print([ (round(t['solver_kernel_ms'],1)) for t in d['per_step']], d['elbo_trace'])"; }
for f in 0 2; do echo "one-wave, twice-flag=$f"; STM_SOLVER_MODE=3 STM_DEBUG_FLAGS=$f run; done
echo two-wave; run
