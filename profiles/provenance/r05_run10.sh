#!/bin/bash
# Run ON the GPU box: the row pass without its walking form (sequential form at 4096^2 fp32 too): GPU suite, every BASELINE size, bench.py as the driver runs it
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r05/run10_tests.txt
tools/sizes.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/run10_sizes.txt
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05/run10_bench20_$i.json; python -c "
import json; j=json.load(open('gpurun_out/r05/run10_bench20_$i.json')); r=j['roofline']; print('driver-style 20 steps:', round(j['value']), 'grids/s', r['kernel'], 'frac', round(r['frac'],3), 'cpu', j['cpu_baseline']['value'] if j['cpu_baseline'] else None)"; done
