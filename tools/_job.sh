export TMPDIR=/tmp
bash tools/profile_r06.sh r06 > gpurun_out/r06_profile.log 2>&1
echo "== c4, three documents per CU in the post step (what a padded matrix would cost)"
for w in 0 3; do STM_POST_MAX_WG_PER_CU=$w timeout 600 python bench.py --config c4 --steps 8 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STM_POST_MAX_WG_PER_CU=$w', round(d['value']), round(d['ms_per_step'],3), {k:round(v.get('avg_launch_ms'),3) for k,v in d['roofline']['kernels'].items()})"; done
tail -5 gpurun_out/r06_profile.log
