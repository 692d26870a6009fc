#!/bin/bash
# Run ON the GPU box: 2048^2 row pass, 16 points per thread + fields one after the other, three workgroups per CU: more sizes and repeats
mkdir -p gpurun_out/r04x
{
echo "== 2048^2 x 4, 200 steps"; N=2048 C=4 STEPS=200 REPS=3 bash tools/ab_4096.sh
echo "== 2048^2 x 2, 300 steps"; N=2048 C=2 STEPS=300 REPS=2 bash tools/ab_4096.sh
echo "== 2048^2 x 1 fp16-stored spectrum, 500 steps"; N=2048 C=1 STEPS=500 REPS=2 EXTRA="--spectrum fp16" bash tools/ab_4096.sh
echo "== 2048^2 x 1, 20 steps after 5 (the driver's way)"; N=2048 C=1 STEPS=20 REPS=3 bash tools/ab_4096.sh
} > gpurun_out/r04x/row_e16_seq_2048.txt 2>&1
cat gpurun_out/r04x/row_e16_seq_2048.txt
