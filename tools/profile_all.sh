#!/bin/bash
# Run ON the GPU box: the round's profile set (kernel trace + PMC passes per size), traffic file, test log, bench lines
R=${R:-r03}
tools/profile_gpu.sh ${R}_prof_1024x4 > /dev/null 2>&1
python tools/make_traffic_json.py gpurun_out/${R}_prof_1024x4 "1024x1024 x 4 cascades" gpurun_out/${R}_traffic.json > /dev/null
tools/profile_gpu.sh ${R}_prof_512x1 --resolution 512 --cascades 1 --steps 2000 --warmup 200 > /dev/null 2>&1
tools/profile_gpu.sh ${R}_prof_2048x1 --resolution 2048 --cascades 1 --steps 300 --warmup 30 > /dev/null 2>&1
tools/profile_gpu.sh ${R}_prof_2048x4 --resolution 2048 --cascades 4 --steps 100 --warmup 10 > /dev/null 2>&1
tools/profile_gpu.sh ${R}_prof_4096 --resolution 4096 --cascades 1 --steps 100 --warmup 10 > /dev/null 2>&1
tools/profile_gpu.sh ${R}_prof_4096h --resolution 4096 --cascades 1 --steps 100 --warmup 10 --spectrum fp16 > /dev/null 2>&1
tools/profile_gpu.sh ${R}_prof_4096h0 --resolution 4096 --cascades 1 --steps 100 --warmup 10 --spectrum fp16h0 > /dev/null 2>&1
tools/profile_gpu.sh ${R}_prof_1024x16 --resolution 1024 --cascades 16 --steps 100 --warmup 10 > /dev/null 2>&1
python -m pytest tests -q -m gpu -v > gpurun_out/${R}_gpu_tests.txt 2>&1
grep -E "passed|failed" gpurun_out/${R}_gpu_tests.txt | tail -1
