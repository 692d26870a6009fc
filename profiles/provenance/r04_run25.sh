#!/bin/bash
# Run ON the GPU box: the shared twiddle table written to the LDS behind the request for the inputs (shipped) against before it (noearly)
mkdir -p gpurun_out/r04x
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
{
echo "== 1024^2 x 4, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=3 bash tools/ab_4096.sh
echo "== 1024^2 x 16, 200 steps"; N=1024 C=16 STEPS=200 REPS=2 bash tools/ab_4096.sh
echo "== 4096^2 fp16-stored spectrum, 200 steps"; N=4096 C=1 STEPS=200 REPS=2 EXTRA="--spectrum fp16" bash tools/ab_4096.sh
echo "== 2048^2 x 1, 500 steps"; N=2048 C=1 STEPS=500 REPS=2 bash tools/ab_4096.sh
echo "== 512^2 x 1, 2000 steps"; N=512 C=1 STEPS=2000 REPS=2 bash tools/ab_4096.sh
echo "== 256^2 x 1, 2000 steps"; N=256 C=1 STEPS=2000 REPS=2 bash tools/ab_4096.sh
echo "== 64^2 x 1, 2000 steps"; N=64 C=1 STEPS=2000 REPS=2 bash tools/ab_4096.sh
} > gpurun_out/r04x/early_request.txt 2>&1
cat gpurun_out/r04x/early_request.txt
