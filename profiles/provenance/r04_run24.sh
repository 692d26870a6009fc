#!/bin/bash
# Run ON the GPU box: block shape of the fp16-stored work spectrum (8-byte points: 8 x 8 blocks make the row pass's stores half lines)
mkdir -p gpurun_out/r04x
{
echo "== 4096^2 fp16-stored spectrum, 200 steps"; N=4096 C=1 STEPS=200 REPS=2 EXTRA="--spectrum fp16" bash tools/ab_4096.sh
echo "== 2048^2 x 1 fp16-stored spectrum, 500 steps"; N=2048 C=1 STEPS=500 REPS=1 EXTRA="--spectrum fp16" bash tools/ab_4096.sh
echo "== 1024^2 x 4 fp16-stored spectrum, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=1 EXTRA="--spectrum fp16" bash tools/ab_4096.sh
for lib in datum_amd/lib/variants/lib_h16c16.so; do echo "-- parity $(basename $lib)"; DATUM_OCEAN_HIP_LIB=$(realpath $lib) timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fp16 or half or spectrum" 2>&1 | tail -3; done
} > gpurun_out/r04x/h16_blocks.txt 2>&1
cat gpurun_out/r04x/h16_blocks.txt
