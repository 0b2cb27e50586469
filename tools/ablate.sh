#!/bin/bash
# builds strutopy_amd/libstm_ablate_<bits>.so for every argument (post_kernel without one phase, stm_post.h); run tools/ablate.py on the GPU box
cd "$(dirname "$0")/.." || exit 1
for b in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics -mllvm -disable-machine-licm \
    -DSTM_ABLATE=$b strutopy_amd/csrc/stm_api.hip -o strutopy_amd/libstm_ablate_$b.so -ldl || exit 1 &
  while [ "$(jobs -r | wc -l)" -ge 4 ]; do sleep 1; done
done
wait
