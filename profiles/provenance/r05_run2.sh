#!/bin/bash
# Run ON the GPU box: the round-5 LDS layouts with unmerged ds_read_b64 (shipped) / with hipcc's ds_read2_b64 pairs (ldsmerged) / round 4's
# paddings (r04): interleaved timings at the BASELINE sizes, then SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel for each build
mkdir -p gpurun_out/r05
{
echo "== 1024^2 x 4, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=3 EXTRA="" tools/ab_4096.sh
echo "== 4096^2 fp16-stored spectrum, 200 steps"; N=4096 C=1 STEPS=200 REPS=3 EXTRA="--spectrum fp16" tools/ab_4096.sh
echo "== 4096^2 fp32, 200 steps"; N=4096 C=1 STEPS=200 REPS=3 EXTRA="" tools/ab_4096.sh
echo "== 2048^2 x 1, 500 steps"; N=2048 C=1 STEPS=500 REPS=3 EXTRA="" tools/ab_4096.sh
echo "== 1024^2 x 16, 200 steps"; N=1024 C=16 STEPS=200 REPS=2 EXTRA="" tools/ab_4096.sh
echo "== 512^2 x 1, 2000 steps"; N=512 C=1 STEPS=2000 REPS=2 EXTRA="" tools/ab_4096.sh
echo "== 1024^2 x 4 fp16-stored spectrum, 1000 steps"; N=1024 C=4 STEPS=1000 REPS=2 EXTRA="--spectrum fp16" tools/ab_4096.sh
} 2>&1 | tee gpurun_out/r05/run2_ab.txt
{
for v in shipped ldsmerged r04; do
  if [ $v = shipped ]; then unset DATUM_OCEAN_HIP_LIB; else export DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_$v.so); fi
  tools/lds/pmc_lds.sh $v --resolution 1024 --cascades 4 --steps 200 --warmup 20
  tools/lds/pmc_lds.sh $v --resolution 4096 --cascades 1 --steps 50 --warmup 5
  tools/lds/pmc_lds.sh $v --resolution 4096 --cascades 1 --steps 50 --warmup 5 --spectrum fp16
  tools/lds/pmc_lds.sh $v --resolution 2048 --cascades 1 --steps 100 --warmup 10
done
} 2>&1 | tee gpurun_out/r05/run2_pmc.txt
