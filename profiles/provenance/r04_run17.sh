#!/bin/bash
# Run ON the GPU box: the row pass's WILD instantiation (phases outside [0, 2 pi): polynomial sin / cos) -- tests, and the normal path against the
# build before it (variant `before`)
mkdir -p gpurun_out/r04o
python -m pytest tests -q -m gpu -x 2>&1 | tail -2
{
echo "== 1024^2 x 4, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=2 bash tools/ab_4096.sh
echo "== 4096^2 fp16"; N=4096 C=1 STEPS=200 REPS=3 EXTRA="--spectrum fp16" bash tools/ab_4096.sh
} > gpurun_out/r04o/ab_wild2.txt 2>&1
cat gpurun_out/r04o/ab_wild2.txt
