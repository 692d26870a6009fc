#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel trace + PMC passes of bench.py; raw output under gpurun_out/$1
# usage: tools/profile_gpu.sh <outdir-name> [bench args...]
set -u
out=gpurun_out/$1; shift
rm -rf $out
mkdir -p $out
export TMPDIR=/tmp
ARGS="--cpu-seconds 0 --no-frame --no-regime $*"   # bench.py defaults (2000 steps, 100 warm-up, 1024x1024 x 4) without the CPU leg and the 64^2 frame
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $ARGS > $out/trace.log 2>&1
i=0
for pmc in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" \
  "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" ; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $out/pmc$i -- python3 bench.py $ARGS > $out/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 tools/summarize_profile.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
