#!/bin/bash
# Run ON the GPU box: more of the seeded differential test (tools/dbg/fuzz.py) on the final build
mkdir -p gpurun_out/r04x
{
for seed in 11 12 13 14 15; do timeout 600 python tools/dbg/fuzz.py 120 $seed 2>&1 | tail -1; done
for seed in 8 9; do FUZZ_SIZES=2048,4096 timeout 900 python tools/dbg/fuzz.py 10 $seed 2>&1 | tail -1; done
} > gpurun_out/r04x/fuzz_more.txt 2>&1
cat gpurun_out/r04x/fuzz_more.txt
