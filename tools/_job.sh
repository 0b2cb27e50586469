export TMPDIR=/tmp
echo "== r05 persistent"; STM_LIB_PATH=$PWD/strutopy_amd/libstm_r05.so timeout 300 python tools/solver_prof.py 100000 10000 50 8 7 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | head -3 | cut -c1-330
echo "== nopersist form, one workgroup per document"; STM_SOLVER_PERSIST=0 STM_LIB_PATH=$PWD/strutopy_amd/libstm_nopersist.so timeout 300 python tools/solver_prof.py 100000 10000 50 8 7 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | head -3 | cut -c1-330
echo "== r05 lib, STM_SOLVER_PERSIST=0"; STM_SOLVER_PERSIST=0 STM_LIB_PATH=$PWD/strutopy_amd/libstm_r05.so timeout 300 python tools/solver_prof.py 100000 10000 50 8 7 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | head -3 | cut -c1-330
echo "== testing lib (c0 in LDS)"; timeout 300 python tools/solver_prof.py 100000 10000 50 8 7 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | head -3 | cut -c1-330
echo "== bitcmp"; timeout 600 python tools/bitcmp.py strutopy_amd/libstm_r05.so strutopy_amd/libstm_hip.so 100000 20 | tail -5
echo "== k512 + new tests"; timeout 900 python -m pytest tests -m gpu -q -x -k "more_than_128 or run_to_run or exchange" 2>&1 | grep -E "passed|failed|rror" | tail -3
