#!/bin/bash
# Run ON the GPU box: random sequences of C-ABI calls against the host model (tools/dbg/api_fuzz.py)
mkdir -p gpurun_out/r04x
{
for seed in 2 3 4 5 6 7; do timeout 900 python tools/dbg/api_fuzz.py 60 $seed 60 2>&1 | tail -1; done
for seed in 8 9; do FUZZ_SIZES=1024,2048 timeout 1200 python tools/dbg/api_fuzz.py 6 $seed 30 2>&1 | tail -1; done
} > gpurun_out/r04x/api_fuzz_more.txt 2>&1
cat gpurun_out/r04x/api_fuzz_more.txt | cut -c1-600
