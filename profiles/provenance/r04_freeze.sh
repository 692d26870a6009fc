#!/bin/bash
# Run ON the GPU box: what has to be measured on the FINAL sources -- the 1024^2 x 4 profile whose traffic file is stamped with the kernel
# sources' hash, the bench lines, the GPU test log
export R=r04
tools/profile_gpu.sh ${R}_prof_1024x4 > /dev/null 2>&1
python tools/make_traffic_json.py gpurun_out/${R}_prof_1024x4 "1024x1024 x 4 cascades" gpurun_out/${R}_traffic.json > /dev/null
cp gpurun_out/${R}_traffic.json profiles/${R}_traffic.json      # (bench.py below quotes it)
cp gpurun_out/${R}_prof_1024x4/summary.txt gpurun_out/${R}_summary_1024x4.txt
cp gpurun_out/${R}_prof_1024x4/trace/*/*kernel_stats.csv gpurun_out/${R}_kernel_stats_1024x4.csv 2>/dev/null
rm -rf gpurun_out/${R}_prof_1024x4
python bench.py > gpurun_out/${R}_bench_1gpu.json 2> /dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_1gpu_20steps.json 2> /dev/null
python bench.py --steps 20 --warmup 5 >> gpurun_out/${R}_bench_1gpu_20steps.json 2> /dev/null
python bench.py --steps 20 --warmup 5 >> gpurun_out/${R}_bench_1gpu_20steps.json 2> /dev/null
python -m pytest tests -q -m gpu -v > gpurun_out/${R}_gpu_tests.txt 2>&1; grep -E "passed|failed" gpurun_out/${R}_gpu_tests.txt | tail -1
cp gpurun_out/parity_table.txt gpurun_out/${R}_parity_table.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
head -8 gpurun_out/${R}_summary_1024x4.txt
