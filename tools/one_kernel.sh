#!/bin/bash
# Registers / scratch / occupancy of ONE solver instantiation in ~20 s instead of the whole library's two minutes:
#   tools/one_kernel.sh "1, 50, false, 2, 0, true" [extra hipcc flags]   (template arguments of stm::solver_kernel; -S output in /tmp/one_kernel.s)
cd "$(dirname "$0")/.." || exit 1
args="${1:-1, 50, false, 2, 0, true}"; shift
cat > /tmp/one_kernel.hip <<EOS
#include <hip/hip_runtime.h>
#include <cstdint>
#include "$PWD/strutopy_amd/csrc/stm_post_common.h"
#include "$PWD/strutopy_amd/csrc/stm_solver.h"
template __global__ void stm::solver_kernel<$args>(stm::SolverParams);
EOS
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -mllvm -disable-machine-licm "$@" \
  --cuda-device-only -S -Rpass-analysis=kernel-resource-usage /tmp/one_kernel.hip -o /tmp/one_kernel.s 2>&1 \
  | grep -E "Function Name|VGPRs:|Spill|Scratch|Occupancy|error" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//'
