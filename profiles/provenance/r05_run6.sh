#!/bin/bash
# Run ON the GPU box: the build without spills (two-step radix-16 twiddles, two-base dispersion addressing): suite, sizes, the 4096^2 fp16 timeline;
# 512^2 x 1 with two-column tiles in the column pass (w2at512: 256 workgroups of 128 threads instead of 128 of 256; 2 x 8 patches)
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r05/run6_tests.txt
tools/sizes.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/run6_sizes.txt
{ echo "== 512^2 x 1, 2000 steps"; N=512 C=1 STEPS=2000 REPS=4 EXTRA="" tools/ab_4096.sh; echo "== 512^2 x 4, 1000 steps"; N=512 C=4 STEPS=1000 REPS=2 EXTRA="" tools/ab_4096.sh; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/run6_ab512.txt
DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_w2at512.so) timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "512" 2>&1 | tail -2
for v in shipped w2at512; do if [ $v = shipped ]; then unset DATUM_OCEAN_HIP_LIB; else export DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_$v.so); fi; echo "gen from 512^2 maps, $v"; python tools/gen_bench.py 512; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/run6_gen512.txt
unset DATUM_OCEAN_HIP_LIB
./tools/dbg/bin/stamps_4096h > gpurun_out/r05/run6_stamps_4096h.txt 2>&1
./tools/dbg/bin/stamps_512 > gpurun_out/r05/run6_stamps_512.txt 2>&1
./tools/dbg/bin/stamps_512_w2 > gpurun_out/r05/run6_stamps_512_w2.txt 2>&1
