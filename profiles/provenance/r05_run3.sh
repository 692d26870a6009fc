#!/bin/bash
# Run ON the GPU box: LDS accesses unpaired through the kernels' target attribute (shipped) against hipcc's pairs (ldsmerged), the column pass
# compiled for four workgroups per CU (colmin4) and round 4 (r04)
mkdir -p gpurun_out/r05
{
echo "== 1024^2 x 4, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=3 EXTRA="" tools/ab_4096.sh
echo "== 1024^2 x 4 fp16-stored spectrum, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=2 EXTRA="--spectrum fp16" tools/ab_4096.sh
echo "== 4096^2 fp16-stored spectrum, 200 steps"; N=4096 C=1 STEPS=200 REPS=3 EXTRA="--spectrum fp16" tools/ab_4096.sh
echo "== 4096^2 fp32, 200 steps"; N=4096 C=1 STEPS=200 REPS=2 EXTRA="" tools/ab_4096.sh
echo "== 2048^2 x 1, 500 steps"; N=2048 C=1 STEPS=500 REPS=2 EXTRA="" tools/ab_4096.sh
echo "== 2048^2 x 4, 300 steps"; N=2048 C=4 STEPS=300 REPS=2 EXTRA="" tools/ab_4096.sh
echo "== 512^2 x 1, 2000 steps"; N=512 C=1 STEPS=2000 REPS=2 EXTRA="" tools/ab_4096.sh
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/run3_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
