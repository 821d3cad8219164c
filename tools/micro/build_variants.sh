#!/bin/bash
# build libmvusba variants into variants/ (git-ignored): name:flags pairs as arguments
set -e
cd "$(dirname "$0")/../../mvus_amd/csrc"
mkdir -p ../../variants
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-variable $flags -shared -o ../../variants/libmvusba_$name.so ba_api.hip -Rpass-analysis=kernel-resource-usage 2> /tmp/variant_$name.log
  echo "$name: $(grep -A8 'k_observationsILb0ELb1' /tmp/variant_$name.log | grep -E 'VGPRs:|ScratchSize|Occupancy' | sed 's/.*remark: *//' | tr '\n' ' ')"
done
