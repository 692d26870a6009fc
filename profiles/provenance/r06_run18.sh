#!/bin/bash
# Run ON the GPU box (round 6): staggered starts -- every other workgroup (by bit SHIFT of its index on its XCD) sleeps SLEEPS x 0.85 us before its first load --
# in the column pass (one round of lock-stepped workgroups at 1024^2 x 4: load, compute, store chip-wide one after the other) and in the row pass
out=gpurun_out/r06_run18; mkdir -p $out
export TMPDIR=/tmp
line() {
  python -c "
import json,sys,os
j=json.loads(sys.stdin.read()); r=j['roofline']; c=j['config']
print(f\"{os.environ.get('VNAME','shipped'):12s} {c['resolution']:5d}^2 x {c['cascades_per_gpu']:2d}  {j['value']:9.0f} grids/s  step {j['ms_per_step']*1e3:8.2f} us  row {r['rowpass']['ms']*1e3:7.2f} us  col {r['colpass']['ms']*1e3:7.2f} us\")"
}
run() { python bench.py --cpu-seconds 0 --no-frame --no-regime "$@" 2>/dev/null | line; }
use() { if [ "$1" = shipped ]; then unset DATUM_OCEAN_HIP_LIB; export VNAME=shipped; else export DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_$1.so); export VNAME=$1; fi; }
{
for rep in 1 2; do
  for v in shipped $(ls datum_amd/lib/variants | sed 's/^lib_//; s/\.so$//'); do use $v
    run --resolution 1024 --cascades 4 --steps 2000 --warmup 100
    run --resolution 2048 --cascades 1 --steps 500 --warmup 50
  done
done
unset DATUM_OCEAN_HIP_LIB
} > $out/stagger.txt 2>&1
cat $out/stagger.txt
