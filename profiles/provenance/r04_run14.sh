#!/bin/bash
mkdir -p gpurun_out/r04m
{
echo "== 4096^2 fp32, 200 steps"; N=4096 C=1 STEPS=200 REPS=2 bash tools/ab_4096.sh
echo "== 4096^2 fp16-stored spectrum, 200 steps"; N=4096 C=1 STEPS=200 REPS=2 EXTRA="--spectrum fp16" bash tools/ab_4096.sh
echo "== 2048^2 x 1, 500 steps"; N=2048 C=1 STEPS=500 REPS=2 bash tools/ab_4096.sh
echo "== 2048^2 x 4, 200 steps"; N=2048 C=4 STEPS=200 REPS=2 bash tools/ab_4096.sh
} > gpurun_out/r04m/ab_col16.txt 2>&1
cat gpurun_out/r04m/ab_col16.txt
for v in col16all; do DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_$v.so) timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "2048 or 4096" 2>&1 | tail -2; done
