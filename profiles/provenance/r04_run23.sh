#!/bin/bash
# Run ON the GPU box: ocean.gen with each workgroup doing several vertically adjacent tiles one after the other (OCEAN_GEN_LOOP)
mkdir -p gpurun_out/r04w
{
for rep in 1 2; do
echo "-- shipped"; python tools/gen_bench.py 64 1024
for lib in datum_amd/lib/variants/lib_genloop*.so; do echo "-- $(basename $lib .so | cut -c5-)"; DATUM_OCEAN_HIP_LIB=$(realpath $lib) python tools/gen_bench.py 64 1024; done
done
for lib in datum_amd/lib/variants/lib_genloop2.so datum_amd/lib/variants/lib_genloop4t128.so; do echo "-- parity $(basename $lib)"; DATUM_OCEAN_HIP_LIB=$(realpath $lib) timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "gen or mesh" 2>&1 | tail -3; done
} > gpurun_out/r04w/gen_loop.txt 2>&1
cat gpurun_out/r04w/gen_loop.txt
