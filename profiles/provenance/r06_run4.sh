#!/bin/bash
# Run ON the GPU box (round 6, fourth call): map-store policy variants of the large grids, the large-grid policy at 1024^2 beyond the cache, in-kernel timeline of 4096^2 fp16
out=gpurun_out/r06_run4; mkdir -p $out
export TMPDIR=/tmp
timeout 200 ./tools/dbg/bin/stamps4096h > $out/stamps4096h.txt 2>&1; head -30 $out/stamps4096h.txt | cut -c1-250
line() {
  python -c "
import json,sys,os
j=json.loads(sys.stdin.read()); r=j['roofline']; c=j['config']
print(f\"{os.environ.get('VNAME','shipped'):14s} {c['resolution']:5d}^2 x {c['cascades_per_gpu']:2d} {'fp16' if 'fp16' in c['workload'] else 'fp32'} group {c['cascades_per_launch']:2d}  {j['value']:9.0f} grids/s  step {j['ms_per_step']*1e3:8.1f} us  row {r['rowpass']['ms']*1e3:7.1f} us  col {r['colpass']['ms']*1e3:7.1f} us  step_frac {r['step_frac']:.3f}  on bytes moved {r['frac_of_peak_on_bytes_moved']['step']:.3f}\")"
}
run() { python bench.py --cpu-seconds 0 --no-frame --no-regime "$@" 2>/dev/null | line; }
{
for rep in 1 2 3; do
  for lib in shipped datum_amd/lib/variants/lib_*.so; do
    if [ "$lib" = shipped ]; then unset DATUM_OCEAN_HIP_LIB; export VNAME=shipped; else export DATUM_OCEAN_HIP_LIB=$(realpath $lib); export VNAME=$(basename $lib .so | cut -c5-); fi
    if [ "$VNAME" != plain1024 ]; then
      run --resolution 4096 --cascades 1 --steps 100 --warmup 10 --spectrum fp16
      run --resolution 4096 --cascades 1 --steps 100 --warmup 10
      run --resolution 2048 --cascades 4 --steps 100 --warmup 10
    fi
    if [ "$VNAME" = shipped ] || [ "$VNAME" = plain1024 ]; then
      run --resolution 1024 --cascades 16 --steps 100 --warmup 10
      run --resolution 1024 --cascades 8 --steps 200 --warmup 20
      run --resolution 1024 --cascades 6 --steps 200 --warmup 20
    fi
  done
done
unset DATUM_OCEAN_HIP_LIB
} > $out/store_policies.txt 2>&1
cat $out/store_policies.txt
