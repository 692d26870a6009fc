#!/bin/bash
# Run ON the GPU box: ocean.gen with every second wave (w1) / workgroup (w2) starting 64 x K clocks late (stK)
mkdir -p gpurun_out/r04x
{
for rep in 1 2; do
echo "-- shipped"; python tools/gen_bench.py 64 1024 2>/dev/null
for lib in datum_amd/lib/variants/lib_st*.so; do echo "-- $(basename $lib .so | cut -c5-)"; DATUM_OCEAN_HIP_LIB=$(realpath $lib) python tools/gen_bench.py 64 1024 2>/dev/null; done
done
} > gpurun_out/r04x/gen_stagger.txt 2>&1
cat gpurun_out/r04x/gen_stagger.txt
