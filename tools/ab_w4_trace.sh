export STM_LIB_PATH=$PWD/strutopy_amd/libstm_w4o3.so
for cfg in "2 16" "2 3" "4 16" "4 2"; do set -- $cfg
STM_POST_BIG2_WAVES=$1 STM_POST_MAX_WG_PER_CU=$2 timeout 600 python bench.py --config c4 --steps 8 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves $1 maxwg $2', round(d['ms_per_step'],2), {k:round(v.get('avg_launch_ms'),2) for k,v in d['roofline']['kernels'].items()}, [repr(x) for x in d.get('elbo_trace')])"
done
