#!/bin/bash
# Run ON the GPU box: row pass with 16 points per thread and the two fields one after the other (the 1024^2 column pass's recipe) at 2048^2 / 1024^2, fp32
mkdir -p gpurun_out/r04x
{
echo "== 2048^2 x 1, 500 steps"; N=2048 C=1 STEPS=500 REPS=2 bash tools/ab_4096.sh
echo "== 2048^2 x 4, 200 steps"; N=2048 C=4 STEPS=200 REPS=1 bash tools/ab_4096.sh
echo "== 1024^2 x 4, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=2 bash tools/ab_4096.sh
echo "== 1024^2 x 16, 200 steps"; N=1024 C=16 STEPS=200 REPS=1 bash tools/ab_4096.sh
for lib in datum_amd/lib/variants/lib_r2048e16seq3.so datum_amd/lib/variants/lib_r1024e16seq6.so; do echo "-- parity $(basename $lib)"; DATUM_OCEAN_HIP_LIB=$(realpath $lib) timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "end_to_end or rowpass or four_cascades or phase_is_bit" 2>&1 | tail -3; done
} > gpurun_out/r04x/row_e16_seq.txt 2>&1
cat gpurun_out/r04x/row_e16_seq.txt
