#!/bin/bash
# Run ON the GPU box: what has to be measured on the FINAL sources -- the GPU suite, the seeded differential runs, the 1024^2 x 4 profile whose traffic
# file is stamped with the kernel sources' hash, the other sizes' profiles, ocean.gen's, the bench lines, smoke()
export R=r05
mkdir -p gpurun_out/$R
python -m pytest tests -q -m gpu -v > gpurun_out/${R}/gpu_tests.txt 2>&1; grep -E "passed|failed" gpurun_out/${R}/gpu_tests.txt | tail -1
cp gpurun_out/parity_table.txt gpurun_out/${R}/parity_table.txt 2>/dev/null
{
for seed in 21 22 23; do timeout 600 python tools/dbg/fuzz.py 120 $seed 2>&1 | tail -1; done
for seed in 24 25; do FUZZ_SIZES=2048,4096 timeout 900 python tools/dbg/fuzz.py 10 $seed 2>&1 | tail -1; done
echo "-- api_fuzz"; for seed in 12 13 14; do timeout 900 python tools/dbg/api_fuzz.py 60 $seed 60 2>&1 | tail -1; done
FUZZ_SIZES=1024,2048 timeout 1200 python tools/dbg/api_fuzz.py 6 15 30 2>&1 | tail -1
echo "-- host_fuzz"; for seed in 12 13 14; do timeout 900 python tools/dbg/host_fuzz.py 60 $seed 150 2>&1 | tail -1; done
} > gpurun_out/${R}/fuzz.txt 2>&1
tools/profile_gpu.sh ${R}/prof_1024x4 > /dev/null 2>&1
python tools/make_traffic_json.py gpurun_out/${R}/prof_1024x4 "1024x1024 x 4 cascades" gpurun_out/${R}/traffic.json > /dev/null
cp gpurun_out/${R}/traffic.json profiles/${R}_traffic.json      # (bench.py below quotes it)
cp gpurun_out/${R}/prof_1024x4/summary.txt gpurun_out/${R}/summary_1024x4.txt
cp gpurun_out/${R}/prof_1024x4/trace/*/*kernel_stats.csv gpurun_out/${R}/kernel_stats_1024x4.csv 2>/dev/null
rm -rf gpurun_out/${R}/prof_1024x4
tools/profile_gpu.sh ${R}/prof_512x1 --resolution 512 --cascades 1 --steps 2000 --warmup 200 > /dev/null 2>&1
tools/profile_gpu.sh ${R}/prof_2048x1 --resolution 2048 --cascades 1 --steps 300 --warmup 30 > /dev/null 2>&1
tools/profile_gpu.sh ${R}/prof_2048x4 --resolution 2048 --cascades 4 --steps 100 --warmup 10 > /dev/null 2>&1
tools/profile_gpu.sh ${R}/prof_4096 --resolution 4096 --cascades 1 --steps 100 --warmup 10 > /dev/null 2>&1
tools/profile_gpu.sh ${R}/prof_4096h --resolution 4096 --cascades 1 --steps 100 --warmup 10 --spectrum fp16 > /dev/null 2>&1
tools/profile_gpu.sh ${R}/prof_1024x16 --resolution 1024 --cascades 16 --steps 200 --warmup 20 > /dev/null 2>&1
tools/profile_gpu.sh ${R}/prof_1024x8 --resolution 1024 --cascades 8 --steps 400 --warmup 40 > /dev/null 2>&1
tools/profile_gen.sh ${R}/prof_gen_64 64 > /dev/null 2>&1
tools/profile_gen.sh ${R}/prof_gen_1024 1024 > /dev/null 2>&1
for d in 512x1 2048x1 2048x4 4096 4096h 1024x16 1024x8 gen_64 gen_1024; do cp gpurun_out/${R}/prof_$d/summary.txt gpurun_out/${R}/summary_$d.txt 2>/dev/null; rm -rf gpurun_out/${R}/prof_$d; done
python bench.py > gpurun_out/${R}/bench_1gpu.json 2> /dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}/bench_1gpu_20steps.json 2> /dev/null
python bench.py --steps 20 --warmup 5 >> gpurun_out/${R}/bench_1gpu_20steps.json 2> /dev/null
python bench.py --steps 20 --warmup 5 >> gpurun_out/${R}/bench_1gpu_20steps.json 2> /dev/null
tools/sizes.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}/sizes.txt
python tools/gen_bench.py 64 256 512 1024 2048 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}/gen_bench.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${R}/smoke.txt
for d in 1024x4 4096h 2048x1; do echo "-- $d"; sed -n 1,6p gpurun_out/${R}/summary_$d.txt; done
