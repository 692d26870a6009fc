#!/bin/bash
# Run ON the GPU box: the pruned build (no knobs, column pass 1024^2 with paired LDS accesses, everything else unpaired): GPU suite, then against round 4
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r05/run4_tests.txt
{
echo "== 1024^2 x 4, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=3 EXTRA="" tools/ab_4096.sh
echo "== 1024^2 x 4 fp16-stored spectrum, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=2 EXTRA="--spectrum fp16" tools/ab_4096.sh
echo "== 1024^2 x 16, 200 steps"; N=1024 C=16 STEPS=200 REPS=2 EXTRA="" tools/ab_4096.sh
echo "== 4096^2 fp16-stored spectrum, 200 steps"; N=4096 C=1 STEPS=200 REPS=3 EXTRA="--spectrum fp16" tools/ab_4096.sh
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/run4_ab.txt
./tools/dbg/bin/stamps_4096h > gpurun_out/r05/run4_stamps_4096h.txt 2>&1
python bench.py 2>/dev/null | tee gpurun_out/r05/run4_bench.json | cut -c1-400
