#!/bin/bash
mkdir -p gpurun_out/r04l
{
for rep in 1 2; do
echo "-- shipped (256 threads per workgroup)"; python tools/gen_bench.py 64 1024
for t in 128 512 1024; do echo "-- $t threads per workgroup"; DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/genvariants/lib_gent$t.so) python tools/gen_bench.py 64 1024; done
done
DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/genvariants/lib_gent512.so) python -m pytest tests/test_gpu_parity.py -m gpu -q -k "gen" 2>&1 | tail -2
DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/genvariants/lib_gent1024.so) python -m pytest tests/test_gpu_parity.py -m gpu -q -k "gen" 2>&1 | tail -2
} > gpurun_out/r04l/gen_threads.txt 2>&1
grep -v libdrm gpurun_out/r04l/gen_threads.txt
