#!/bin/bash
# Run ON the GPU box: full GPU suite on the shipped build (hardware sin / cos in the row pass); store-policy sweep of the step kernels on the 24-byte layout;
# ocean.gen levers: store policy of the vertex stream, split launch
mkdir -p gpurun_out/r04e
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r04e/tests.log 2>&1; tail -4 gpurun_out/r04e/tests.log
cp gpurun_out/parity_table.txt gpurun_out/r04e/parity_table.txt
{
echo "== 1024^2 x 4, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=2 bash tools/ab_4096.sh
echo "== 1024^2 x 16, 200 steps"; N=1024 C=16 STEPS=200 REPS=2 bash tools/ab_4096.sh
echo "== 4096^2 fp16-stored spectrum, 200 steps"; N=4096 C=1 STEPS=200 REPS=2 EXTRA="--spectrum fp16" bash tools/ab_4096.sh
echo "== 2048^2 x 1, 500 steps"; N=2048 C=1 STEPS=500 REPS=1 bash tools/ab_4096.sh
} > gpurun_out/r04e/ab_step.txt 2>&1
cat gpurun_out/r04e/ab_step.txt
frame() { python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-regime 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   bench: gen %.2f us (1024^2 maps), reference frame n64 %.2f us (displace only %.2f)' % (j['gen']['ms']*1e3, j['reference_frame_n64']['us_per_frame'], j['reference_frame_n64']['us_displace_only']))"; }
{
for rep in 1 2; do
echo "-- shipped (plain stores, one launch)"; python tools/gen_bench.py 64 1024; frame
echo "-- split launch, two streams (DATUM_OCEAN_GEN_SPLIT_ROWS=512)"; DATUM_OCEAN_GEN_SPLIT_ROWS=512 python tools/gen_bench.py 64 1024; DATUM_OCEAN_GEN_SPLIT_ROWS=512 frame
for v in 1 2 16 17 18; do echo "-- vertex stores with aux $v"; DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/genvariants/lib_genst$v.so) python tools/gen_bench.py 64 1024; done
done
} > gpurun_out/r04e/gen_levers.txt 2>&1
cat gpurun_out/r04e/gen_levers.txt
