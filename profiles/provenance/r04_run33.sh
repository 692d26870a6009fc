#!/bin/bash
# Run ON the GPU box: random sequences of calls on the C++ host shim against the host model (tools/dbg/host_fuzz.py)
mkdir -p gpurun_out/r04x
{
for seed in 2 3 4 5 6 7 8 9; do timeout 900 python tools/dbg/host_fuzz.py 60 $seed 150 2>&1 | tail -1; done
} > gpurun_out/r04x/host_fuzz_more.txt 2>&1
cat gpurun_out/r04x/host_fuzz_more.txt | cut -c1-900
