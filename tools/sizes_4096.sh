#!/bin/bash
# Run ON the GPU box: 4096^2 only, shipped library and every variant
run() {
  python bench.py --cpu-seconds 0 --no-check --no-frame "$@" | python -c "
import json,sys,os
j=json.loads(sys.stdin.read()); r=j['roofline']; c=j['config']
print(f\"{os.environ.get('VNAME','shipped'):16s} {c['resolution']:5d}^2 x {c['cascades_per_gpu']:2d} {'fp16' if 'fp16' in c['workload'] else 'fp32'}  {j['value']:9.0f} grids/s  step {j['ms_per_step']*1e3:8.1f} us  row {r['rowpass']['ms']*1e3:7.1f} us  col {r['colpass']['ms']*1e3:7.1f} us  step_frac {r['step_frac']:.3f}\")"
}
for lib in shipped datum_amd/lib/variants/lib_*.so; do
  if [ "$lib" = shipped ]; then unset DATUM_OCEAN_HIP_LIB; export VNAME=shipped; else [ -f "$lib" ] || continue; export DATUM_OCEAN_HIP_LIB=$(realpath $lib); export VNAME=$(basename $lib .so | cut -c5-); fi
  run --resolution 4096 --cascades 1 --steps 60 --warmup 6
  [ -z "${FP16:-}" ] || run --resolution 4096 --cascades 1 --steps 60 --warmup 6 --spectrum fp16
done
