#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel trace + PMC passes of the beyond-Infinity-Cache workload (1024^2 x 16) and of 1024^2 x 8
export R=r04
tools/profile_gpu.sh ${R}_prof_1024x16 --resolution 1024 --cascades 16 --steps 200 --warmup 20 > /dev/null 2>&1
tools/profile_gpu.sh ${R}_prof_1024x8 --resolution 1024 --cascades 8 --steps 400 --warmup 40 > /dev/null 2>&1
for d in 1024x16 1024x8; do cp gpurun_out/${R}_prof_$d/summary.txt gpurun_out/${R}_summary_$d.txt; done
rm -rf gpurun_out/${R}_prof_*
head -6 gpurun_out/${R}_summary_1024x16.txt; grep -E "FETCH_SIZE|WRITE_SIZE" gpurun_out/${R}_summary_1024x16.txt
