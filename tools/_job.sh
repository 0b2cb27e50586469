export TMPDIR=/tmp
python tools/dump_long_state.py 300 2>&1 | tail -2
for lib in libstm_hip.so libstm_nopersist.so; do echo "== $lib"; STM_LIB_PATH=$PWD/strutopy_amd/$lib timeout 300 python tools/solver_prof.py 100000 10000 50 8 7 2>&1 | head -12 | cut -c1-400; done
