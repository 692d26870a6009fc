#!/bin/bash
# Run ON the GPU box: the 4096^2 fp16 row pass persistent (two workgroups per CU walk four pairs each, no prefetch) against one workgroup per pair
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "4096 or fp16 or random_parameters or power_of_two" 2>&1 | tail -3
{
echo "== 4096^2 fp16-stored spectrum, 200 steps"; N=4096 C=1 STEPS=200 REPS=4 EXTRA="--spectrum fp16" tools/ab_4096.sh
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/run5_ab.txt
./tools/dbg/bin/stamps_4096h > gpurun_out/r05/run5_stamps_4096h.txt 2>&1
