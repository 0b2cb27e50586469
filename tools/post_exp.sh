run() { python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print([ (round(t['solver_kernel_ms'],1), round(t['post_kernel_ms'],1)) for t in d['per_step']])"; }
echo wpe2; run
for f in 2 4 8; do echo "flags=$f"; STM_POST_DEBUG=$f run; done
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo wpe1; cp strutopy_amd/libstm_hip_wpe1.so strutopy_amd/libstm_hip.so; run; echo wpe1-4percu; STM_POST_MAX_WG_PER_CU=4 run
