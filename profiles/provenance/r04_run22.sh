#!/bin/bash
# Run ON the GPU box: cache policies of the loads -- the row pass's phase (read once) and h0 (read twice), the column pass's spectrum
mkdir -p gpurun_out/r04r
{
echo "== 1024^2 x 4, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=2 bash tools/ab_4096.sh
echo "== 1024^2 x 16, 200 steps"; N=1024 C=16 STEPS=200 REPS=2 bash tools/ab_4096.sh
echo "== 2048^2 x 1"; N=2048 C=1 STEPS=500 REPS=1 bash tools/ab_4096.sh
} > gpurun_out/r04r/ab_loads.txt 2>&1
cat gpurun_out/r04r/ab_loads.txt
