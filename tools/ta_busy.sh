#!/bin/bash
# Run ON the GPU box: texture-path and ALU busy counters of ocean.gen (tools/gen_bench.py 64) for the shipped library and every variant
export TMPDIR=/tmp
for lib in shipped datum_amd/lib/variants/lib_*.so; do
  if [ "$lib" = shipped ]; then unset DATUM_OCEAN_HIP_LIB; name=shipped; else [ -f "$lib" ] || continue; export DATUM_OCEAN_HIP_LIB=$(realpath $lib); name=$(basename $lib .so | cut -c5-); fi
  for pmc in "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
    rm -rf /tmp/ta_$name; rocprofv3 --pmc $pmc --output-format csv -d /tmp/ta_$name -- python3 tools/gen_bench.py ${N:-64} > /dev/null 2>&1
    python3 - $name /tmp/ta_$name <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[2]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gen' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[1], ' '.join(f"{k}={sum(v)/len(v):.0f}" for k,v in sorted(acc.items())))
PY
  done
done
