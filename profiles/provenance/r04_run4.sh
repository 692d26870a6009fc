#!/bin/bash
# Run ON the GPU box: row pass with the packed prologue (shipped) against the scalar one (rowscalar) and hardware sin / cos (hwsin)
mkdir -p gpurun_out/r04d
timeout 600 python -m pytest tests/test_gpu_host_shim.py -m gpu -q > gpurun_out/r04d/tests_host.log 2>&1; tail -3 gpurun_out/r04d/tests_host.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r04d/tests_parity_packed.log 2>&1; tail -3 gpurun_out/r04d/tests_parity_packed.log
cp gpurun_out/parity_table.txt gpurun_out/r04d/parity_packed.txt
DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_hwsin.so) timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q > gpurun_out/r04d/tests_parity_hwsin.log 2>&1; tail -5 gpurun_out/r04d/tests_parity_hwsin.log
cp gpurun_out/parity_table.txt gpurun_out/r04d/parity_hwsin.txt
{
echo "== 1024^2 x 4, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=3 bash tools/ab_4096.sh
echo "== 1024^2 x 4, 20 steps (the driver's way)"; N=1024 C=4 STEPS=20 REPS=3 bash tools/ab_4096.sh
echo "== 1024^2 x 16, 200 steps"; N=1024 C=16 STEPS=200 REPS=2 bash tools/ab_4096.sh
echo "== 4096^2 fp16-stored spectrum, 200 steps"; N=4096 C=1 STEPS=200 REPS=3 EXTRA="--spectrum fp16" bash tools/ab_4096.sh
echo "== 4096^2 fp32, 200 steps"; N=4096 C=1 STEPS=200 REPS=2 bash tools/ab_4096.sh
echo "== 2048^2 x 1, 500 steps"; N=2048 C=1 STEPS=500 REPS=2 bash tools/ab_4096.sh
echo "== 512^2 x 1, 2000 steps"; N=512 C=1 STEPS=2000 REPS=2 bash tools/ab_4096.sh
} > gpurun_out/r04d/ab_rowpass.txt 2>&1
cat gpurun_out/r04d/ab_rowpass.txt
