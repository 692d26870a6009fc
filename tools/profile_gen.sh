#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel trace + PMC passes of tools/gen_bench.py; raw output under gpurun_out/$1
# usage: tools/profile_gen.sh <outdir-name> [N ...]
set -u
out=gpurun_out/$1; shift
rm -rf $out
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/gen_bench.py $* > $out/trace.log 2>&1
i=0
for pmc in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $out/pmc$i -- python3 tools/gen_bench.py $* > $out/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 tools/summarize_profile.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
