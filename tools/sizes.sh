#!/bin/bash
# Run ON the GPU box: bench.py at the other BASELINE sizes (event times per kernel, step fraction on algorithmic bytes)
# usage: tools/sizes.sh [lib.so]    (lib: a variant of the module; default the shipped one)
[ -n "${1:-}" ] && export DATUM_OCEAN_HIP_LIB=$(realpath $1)
run() {
  python bench.py --cpu-seconds 0 --no-regime "$@" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; c=j['config']
print(f\"{c['resolution']:5d}^2 x {c['cascades_per_gpu']:2d} {'fp16' if 'fp16' in c['workload'] else 'fp32'}  {j['value']:9.0f} grids/s  step {j['ms_per_step']*1e3:8.1f} us  row {r['rowpass']['ms']*1e3:7.1f} us  col {r['colpass']['ms']*1e3:7.1f} us  step_frac {r['step_frac']:.3f}  dominant {r['kernel']} frac {r['frac']:.3f}\")"
}
run --resolution 1024 --cascades 4 --steps 1000 --warmup 100
run --resolution 512 --cascades 1 --steps 2000 --warmup 200
run --resolution 1024 --cascades 8 --steps 500 --warmup 50
run --resolution 2048 --cascades 1 --steps 500 --warmup 50
run --resolution 2048 --cascades 4 --steps 300 --warmup 30
run --resolution 1024 --cascades 16 --steps 300 --warmup 30
run --resolution 4096 --cascades 1 --steps 200 --warmup 20
run --resolution 4096 --cascades 1 --steps 200 --warmup 20 --spectrum fp16
run --resolution 1024 --cascades 4 --steps 1000 --warmup 100 --spectrum fp16
