#!/bin/bash
# round 6's confirmation on the GPU box: the -m gpu suite, the fit bit for bit against round 5's build (strutopy_amd/libstm_r05.so, when
# present), the timeline of one EM iteration, C3's per-GPU share and the driver's bench
export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | tail -5
if [ -f strutopy_amd/libstm_r05.so ]; then echo "== bitcmp"; timeout 600 python tools/bitcmp.py strutopy_amd/libstm_r05.so strutopy_amd/libstm_hip.so 100000 20 | tail -8; fi
bash tools/trace_timeline.sh
echo "== small"; timeout 200 python bench.py --docs 12500 --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v.get('avg_launch_ms') for k,v in d['roofline']['kernels'].items()})"
echo "== default"; for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v.get('avg_launch_ms') for k,v in d['roofline']['kernels'].items()})"; done
echo "== solver cycles"; timeout 300 python tools/solver_prof.py 100000 10000 50 8 7 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | head -1 | cut -c1-330
