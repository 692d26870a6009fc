#!/bin/bash
# Run ON the GPU box: the round's profile set on the final kernels -- rocprofv3 kernel trace + PMC passes per size, traffic file, ocean.gen profile,
# bench lines (default and the driver's way), sizes, GPU test log
export R=r04
mkdir -p gpurun_out
bash tools/profile_all.sh
bash tools/profile_gen.sh ${R}_prof_gen_64 64 > /dev/null 2>&1
bash tools/profile_gen.sh ${R}_prof_gen_1024 1024 > /dev/null 2>&1
python bench.py > gpurun_out/${R}_bench_1gpu.json 2> gpurun_out/${R}_bench_1gpu.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_1gpu_20steps.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-regime --no-frame >> gpurun_out/${R}_bench_1gpu_20steps.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-regime --no-frame >> gpurun_out/${R}_bench_1gpu_20steps.json 2> /dev/null
bash tools/sizes.sh > gpurun_out/${R}_sizes.txt 2>&1
for d in 1024x4 512x1 2048x1 2048x4 4096 4096h gen_64 gen_1024; do cp gpurun_out/${R}_prof_$d/summary.txt gpurun_out/${R}_summary_$d.txt; done
cp gpurun_out/${R}_prof_1024x4/trace/*/*kernel_stats.csv gpurun_out/${R}_kernel_stats_1024x4.csv 2>/dev/null
cp gpurun_out/parity_table.txt gpurun_out/${R}_parity_table.txt
rm -rf gpurun_out/${R}_prof_*
cat gpurun_out/${R}_sizes.txt; head -30 gpurun_out/${R}_summary_1024x4.txt
