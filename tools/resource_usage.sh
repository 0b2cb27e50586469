#!/bin/bash
# registers / scratch / occupancy of every kernel whose mangled name matches $1 (default: all), with the build's flags
cd "$(dirname "$0")/.." && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics -mllvm -disable-machine-licm \
  -Rpass-analysis=kernel-resource-usage strutopy_amd/csrc/stm_api.hip -o /tmp/stm_ru.so -ldl 2>&1 \
  | grep -A12 "Function Name: .*${1:-.}" | grep -E "Name|VGPRs:|Spill|Scratch|Occupancy" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//'
