#!/bin/bash
mkdir -p gpurun_out/r04j
{
echo "== 4096^2 fp16-stored spectrum, 200 steps"; N=4096 C=1 STEPS=200 REPS=3 EXTRA="--spectrum fp16" bash tools/ab_4096.sh
echo "== 4096^2 fp32, 200 steps"; N=4096 C=1 STEPS=200 REPS=2 bash tools/ab_4096.sh
} > gpurun_out/r04j/ab_4096.txt 2>&1
cat gpurun_out/r04j/ab_4096.txt
