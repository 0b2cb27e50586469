#!/bin/bash
# tools/ab_build.sh <name> [hipcc flags ...]  ->  strutopy_amd/libstm_<name>.so with the build's flags plus the given ones (A/B builds
# for tools/bitcmp.py / tools/ab_libs.sh; *.so is git-ignored and travels to the GPU box).  Built with -DSTM_TESTING: the profilers
# (tools/solver_prof.py ...) need the debug switches, which the product build does not have.
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics -mllvm -disable-machine-licm -DSTM_TESTING "$@" \
  strutopy_amd/csrc/stm_api.hip -o strutopy_amd/libstm_$name.so -ldl
