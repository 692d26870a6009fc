#!/bin/bash
# Run ON the GPU box: the remaining rocprofv3 summaries on the final sources (512x1, 4096, 4096h, 1024x16, 1024x8, gen from 64^2 and 1024^2 maps)
export R=r04
tools/profile_gpu.sh ${R}_prof_512x1 --resolution 512 --cascades 1 --steps 2000 --warmup 200 > /dev/null 2>&1
tools/profile_gpu.sh ${R}_prof_4096 --resolution 4096 --cascades 1 --steps 100 --warmup 10 > /dev/null 2>&1
tools/profile_gpu.sh ${R}_prof_4096h --resolution 4096 --cascades 1 --steps 100 --warmup 10 --spectrum fp16 > /dev/null 2>&1
tools/profile_gpu.sh ${R}_prof_1024x16 --resolution 1024 --cascades 16 --steps 200 --warmup 20 > /dev/null 2>&1
tools/profile_gpu.sh ${R}_prof_1024x8 --resolution 1024 --cascades 8 --steps 400 --warmup 40 > /dev/null 2>&1
tools/profile_gen.sh ${R}_prof_gen_64 64 > /dev/null 2>&1
tools/profile_gen.sh ${R}_prof_gen_1024 1024 > /dev/null 2>&1
for d in 512x1 4096 4096h 1024x16 1024x8 gen_64 gen_1024; do cp gpurun_out/${R}_prof_$d/summary.txt gpurun_out/${R}_summary_$d.txt 2>/dev/null; rm -rf gpurun_out/${R}_prof_$d; done
for d in 512x1 4096 4096h 1024x16 1024x8 gen_64 gen_1024; do echo "-- $d"; sed -n 2,4p gpurun_out/${R}_summary_$d.txt; done
