export TMPDIR=/tmp
echo "== default bench (live counters)"; time (timeout 900 python bench.py > gpurun_out/live_bench.json 2> gpurun_out/live_bench.err); grep "live" gpurun_out/live_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/live_bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['traffic']); print(r['traffic_source'][:90]); print(r['compute_source'][:90])
for k,v in r['kernels'].items(): print(k, v['traffic'], v['avg_launch_ms'], v['compute'])
PY
echo "== driver's command"; time (timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/live_bench20.json 2> gpurun_out/live_bench20.err); grep "live" gpurun_out/live_bench20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/live_bench20.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['traffic'])
for k,v in r['kernels'].items(): print(k, v['traffic'], v['avg_launch_ms'], v['compute'])
PY
echo "== c4"; time (timeout 900 python bench.py --config c4 --steps 8 --warmup 2 > gpurun_out/live_c4.json 2> gpurun_out/live_c4.err); grep "live" gpurun_out/live_c4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/live_c4.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['traffic'])
for k,v in r['kernels'].items(): print(k, v['traffic'], v['avg_launch_ms'], v['compute'])
PY
