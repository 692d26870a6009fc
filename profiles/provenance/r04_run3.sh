#!/bin/bash
# Run ON the GPU box: the 24-byte map layout against round 3's 32-byte layout (variant map32), interleaved; the three host-shim tests; hw sin accuracy
mkdir -p gpurun_out/r04c
timeout 600 python -m pytest tests/test_gpu_host_shim.py -m gpu -q > gpurun_out/r04c/tests_host.log 2>&1; tail -3 gpurun_out/r04c/tests_host.log
./tools/dbg/bin/hwsin > gpurun_out/r04c/hwsin.txt 2>&1; cat gpurun_out/r04c/hwsin.txt
{
echo "== 1024^2 x 4, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=3 bash tools/ab_4096.sh
echo "== 1024^2 x 4, 20 steps (the driver's way)"; N=1024 C=4 STEPS=20 REPS=4 bash tools/ab_4096.sh
echo "== 1024^2 x 16, 200 steps"; N=1024 C=16 STEPS=200 REPS=3 bash tools/ab_4096.sh
echo "== 4096^2 fp16-stored spectrum, 200 steps"; N=4096 C=1 STEPS=200 REPS=3 EXTRA="--spectrum fp16" bash tools/ab_4096.sh
echo "== 4096^2 fp32, 200 steps"; N=4096 C=1 STEPS=200 REPS=2 bash tools/ab_4096.sh
echo "== 2048^2 x 1, 500 steps"; N=2048 C=1 STEPS=500 REPS=2 bash tools/ab_4096.sh
echo "== 2048^2 x 4, 200 steps"; N=2048 C=4 STEPS=200 REPS=2 bash tools/ab_4096.sh
echo "== 512^2 x 1, 2000 steps"; N=512 C=1 STEPS=2000 REPS=2 bash tools/ab_4096.sh
echo "== ocean.gen, 1024^2 mesh"; for rep in 1 2; do echo "24-byte layout:"; python tools/gen_bench.py 64 256 1024 2048; echo "32-byte layout:"; DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_map32.so) python tools/gen_bench.py 64 256 1024 2048; done
} > gpurun_out/r04c/ab_layout.txt 2>&1
cat gpurun_out/r04c/ab_layout.txt
