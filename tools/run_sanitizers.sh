#!/bin/bash
# ASan + UBSan over the host-side C (oracle/stm_oracle.c, strutopy_amd/csrc/packbow.c): `make -C oracle asan` builds both into
# oracle/_asan/, this script swaps them in for the regular builds, runs the CPU tests that drive them (tests/test_host_logic.py,
# tests/test_oracle_golden.py) with the sanitizer runtime preloaded, and puts the regular builds back.  CPU only (the GPU pool has no
# sanitizer support); exit status = pytest's.  Any report fails the run (-fno-sanitize-recover, ASAN's default abort on error).
cd "$(dirname "$0")/.." || exit 1
make -s -C oracle asan || exit 1
ext=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
pb=strutopy_amd/_packbow$ext; orc=oracle/libstm_oracle.so
[ -f $orc ] || make -s -C oracle
[ -f $pb ] || python3 -c "import __graft_entry__ as g" 2>/dev/null
restore() { [ -f $orc.regular ] && mv -f $orc.regular $orc; [ -f $pb.regular ] && mv -f $pb.regular $pb; }
trap restore EXIT
cp -p $orc $orc.regular && cp -f oracle/_asan/libstm_oracle.so $orc
[ -f $pb ] && cp -p $pb $pb.regular; cp -f oracle/_asan/_packbow$ext $pb
asan=$(gcc -print-file-name=libasan.so); ubsan=$(gcc -print-file-name=libubsan.so)
# (Python itself is not instrumented: leak checking would report the interpreter's own allocations)
LD_PRELOAD="$asan $ubsan" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 OMP_NUM_THREADS=4 \
  timeout 1800 python3 -m pytest tests/test_host_logic.py tests/test_oracle_golden.py -x -q -p no:cacheprovider "$@"
