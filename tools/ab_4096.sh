#!/bin/bash
# Run ON the GPU box: A/B of the shipped library against every variant at one size, REPS interleaved repeats of STEPS steps
# usage: [N=4096] [C=1] [STEPS=200] [REPS=3] [EXTRA="--spectrum fp16"] tools/ab_4096.sh
N=${N:-4096}; C=${C:-1}; STEPS=${STEPS:-200}; REPS=${REPS:-3}
run() {
  python bench.py --cpu-seconds 0 --no-check --no-frame --no-regime --resolution $N --cascades $C --steps $STEPS --warmup 20 $EXTRA 2>/dev/null | python -c "
import json,sys,os
j=json.loads(sys.stdin.read()); r=j['roofline']
print(f\"{os.environ.get('VNAME','shipped'):16s} {j['value']:9.0f} grids/s  step {j['ms_per_step']*1e3:8.1f} us  row {r['rowpass']['ms']*1e3:7.1f} us  col {r['colpass']['ms']*1e3:7.1f} us  step_frac {r['step_frac']:.3f}\")"
}
for rep in $(seq $REPS); do
  for lib in shipped datum_amd/lib/variants/lib_*.so; do
    if [ "$lib" = shipped ]; then unset DATUM_OCEAN_HIP_LIB; export VNAME=shipped; else [ -f "$lib" ] || continue; export DATUM_OCEAN_HIP_LIB=$(realpath $lib); export VNAME=$(basename $lib .so | cut -c5-); fi
    run
  done
done
