#!/bin/bash
# Run ON the GPU box: ocean.gen timing (tools/gen_bench.py) for the shipped library and every variant, REPS interleaved repeats
REPS=${REPS:-2}; SIZES=${SIZES:-"64 1024"}
for rep in $(seq $REPS); do
  for lib in shipped datum_amd/lib/variants/lib_*.so; do
    if [ "$lib" = shipped ]; then unset DATUM_OCEAN_HIP_LIB; name=shipped; else [ -f "$lib" ] || continue; export DATUM_OCEAN_HIP_LIB=$(realpath $lib); name=$(basename $lib .so | cut -c5-); fi
    python tools/gen_bench.py $SIZES 2>/dev/null | sed "s/^/$name  /"
  done
done
