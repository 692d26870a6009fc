#!/bin/bash
# Run ON the GPU box: the new 2048^2 row pass as shipped against the old form (r2048old) and with the packed prologue (r2048packed); parity on the shipped build
mkdir -p gpurun_out/r04x
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_shim.py -m gpu -q -x 2>&1 | tail -3
{
echo "== 2048^2 x 1, 500 steps"; N=2048 C=1 STEPS=500 REPS=3 bash tools/ab_4096.sh
echo "== 2048^2 x 4, 200 steps"; N=2048 C=4 STEPS=200 REPS=2 bash tools/ab_4096.sh
echo "== 2048^2 x 1 fp16-stored spectrum, 500 steps"; N=2048 C=1 STEPS=500 REPS=2 EXTRA="--spectrum fp16" bash tools/ab_4096.sh
} > gpurun_out/r04x/row_2048_adopted.txt 2>&1
cat gpurun_out/r04x/row_2048_adopted.txt
