#!/bin/bash
# Run ON the GPU box: after the last kernel change of the round (wave-local exchanges at 512^2) -- the GPU suite, the 1024^2 x 4 profile whose traffic
# file is stamped with the kernel code's hash, the 512^2 profile, the bench lines, the sizes, smoke()
export R=r05
mkdir -p gpurun_out/$R
python -m pytest tests -q -m gpu -v > gpurun_out/${R}/gpu_tests.txt 2>&1; grep -E "passed|failed" gpurun_out/${R}/gpu_tests.txt | tail -1
cp gpurun_out/parity_table.txt gpurun_out/${R}/parity_table.txt 2>/dev/null
tools/profile_gpu.sh ${R}/prof_1024x4 > /dev/null 2>&1
python tools/make_traffic_json.py gpurun_out/${R}/prof_1024x4 "1024x1024 x 4 cascades" gpurun_out/${R}/traffic.json > /dev/null
cp gpurun_out/${R}/traffic.json profiles/${R}_traffic.json      # (bench.py below quotes it)
cp gpurun_out/${R}/prof_1024x4/summary.txt gpurun_out/${R}/summary_1024x4.txt
cp gpurun_out/${R}/prof_1024x4/trace/*/*kernel_stats.csv gpurun_out/${R}/kernel_stats_1024x4.csv 2>/dev/null
rm -rf gpurun_out/${R}/prof_1024x4
tools/profile_gpu.sh ${R}/prof_512x1 --resolution 512 --cascades 1 --steps 2000 --warmup 200 > /dev/null 2>&1
cp gpurun_out/${R}/prof_512x1/summary.txt gpurun_out/${R}/summary_512x1.txt; rm -rf gpurun_out/${R}/prof_512x1
python bench.py > gpurun_out/${R}/bench_1gpu.json 2> /dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}/bench_1gpu_20steps.json 2> /dev/null
python bench.py --steps 20 --warmup 5 >> gpurun_out/${R}/bench_1gpu_20steps.json 2> /dev/null
python bench.py --steps 20 --warmup 5 >> gpurun_out/${R}/bench_1gpu_20steps.json 2> /dev/null
tools/sizes.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}/sizes.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${R}/smoke.txt
sed -n 1,4p gpurun_out/${R}/summary_1024x4.txt
