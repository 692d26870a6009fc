#!/bin/bash
# Run ON the GPU box: freeze on the sources with the new 2048^2 row pass + the 2048 profiles and the sizes table
bash tools/r04_freeze.sh
export R=r04
tools/profile_gpu.sh ${R}_prof_2048x1 --resolution 2048 --cascades 1 --steps 300 --warmup 30 > /dev/null 2>&1
tools/profile_gpu.sh ${R}_prof_2048x4 --resolution 2048 --cascades 4 --steps 100 --warmup 10 > /dev/null 2>&1
for s in 2048x1 2048x4; do cp gpurun_out/${R}_prof_$s/summary.txt gpurun_out/${R}_summary_$s.txt; rm -rf gpurun_out/${R}_prof_$s; done
bash tools/sizes.sh > gpurun_out/${R}_sizes.txt 2>/dev/null; cat gpurun_out/${R}_sizes.txt
