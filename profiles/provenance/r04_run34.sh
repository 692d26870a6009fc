#!/bin/bash
# Run ON the GPU box: the 2048^2 recipe at 1024^2 with two row pairs per 256-thread workgroup; 4096^2 fp16 row pass without its spill (one workgroup per CU)
mkdir -p gpurun_out/r04x
{
echo "== 1024^2 x 4, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=2 bash tools/ab_4096.sh
echo "== 1024^2 x 16, 200 steps"; N=1024 C=16 STEPS=200 REPS=1 bash tools/ab_4096.sh
echo "== 4096^2 fp16-stored spectrum, 200 steps"; N=4096 C=1 STEPS=200 REPS=2 EXTRA="--spectrum fp16" bash tools/ab_4096.sh
} > gpurun_out/r04x/row_1024_pairs.txt 2>&1
cat gpurun_out/r04x/row_1024_pairs.txt
