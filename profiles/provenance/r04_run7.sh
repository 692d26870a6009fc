#!/bin/bash
# Run ON the GPU box: the row pass's fields one after the other at 1024^2 compiled for 5 / 6 / 8 workgroups per CU
mkdir -p gpurun_out/r04g
{
echo "== 1024^2 x 4, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=2 bash tools/ab_4096.sh
echo "== 1024^2 x 16, 200 steps"; N=1024 C=16 STEPS=200 REPS=1 bash tools/ab_4096.sh
} > gpurun_out/r04g/ab_rowseq.txt 2>&1
cat gpurun_out/r04g/ab_rowseq.txt
