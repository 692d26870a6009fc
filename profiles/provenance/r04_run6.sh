#!/bin/bash
# Run ON the GPU box: row-pass structure variants at 1024^2 (walking with 3 / 4 workgroups per CU, fields one after the other), 4 points per thread at 512^2
mkdir -p gpurun_out/r04f
{
echo "== 1024^2 x 4, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=2 bash tools/ab_4096.sh
echo "== 1024^2 x 16, 200 steps"; N=1024 C=16 STEPS=200 REPS=1 bash tools/ab_4096.sh
echo "== 2048^2 x 1, 500 steps"; N=2048 C=1 STEPS=500 REPS=1 bash tools/ab_4096.sh
echo "== 512^2 x 1, 2000 steps"; N=512 C=1 STEPS=2000 REPS=2 bash tools/ab_4096.sh
} > gpurun_out/r04f/ab_structure.txt 2>&1
cat gpurun_out/r04f/ab_structure.txt
for v in row512e4 col512e4 both512e4; do DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_$v.so) timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "512 or random" 2>&1 | tail -2; done
