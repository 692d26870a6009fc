#!/bin/bash
# Builds tuning variants of the HIP module into datum_amd/lib/variants/ (they travel with gpurun, git ignores *.so).
# usage: tools/build_variants.sh name "flags" [name "flags" ...]
set -e
cd "$(dirname "$0")/.."
mkdir -p datum_amd/lib/variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $flags \
     -shared -o datum_amd/lib/variants/lib_$name.so datum_amd/csrc/ocean_capi.hip &
done
wait
ls -la datum_amd/lib/variants/
