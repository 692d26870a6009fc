#!/bin/bash
# Run ON the GPU box: timing-only ablations of the row pass at 512^2 x 1 (what is left of its 6.3 us without loads / stores / transforms / all three)
mkdir -p gpurun_out/r04x
{
echo "== 512^2 x 1, 2000 steps"; N=512 C=1 STEPS=2000 REPS=2 bash tools/ab_4096.sh
echo "== 64^2 x 1, 2000 steps"; N=64 C=1 STEPS=2000 REPS=1 bash tools/ab_4096.sh
} > gpurun_out/r04x/ablate_512.txt 2>&1
cat gpurun_out/r04x/ablate_512.txt
