#!/bin/bash
mkdir -p gpurun_out/r04h
{
for shape in "64 1024" "64 512" "64 992"; do
rm -f /tmp/a.npy
DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/genvariants/lib_genst0.so) python tools/dbg/gen_compare.py $shape /tmp/a.npy
python tools/dbg/gen_compare.py $shape /tmp/a.npy
DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/genvariants/lib_genst17.so) python tools/dbg/gen_compare.py $shape /tmp/a.npy
done
} > gpurun_out/r04h/compare.txt 2>&1
grep -v libdrm gpurun_out/r04h/compare.txt
