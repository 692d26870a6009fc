#!/bin/bash
# Run ON the GPU box (round 6, sixth call): the 2048^2 column pass without its spill (fenced last-pass tasks) against round 5's; the whole GPU suite on the new policies
out=gpurun_out/r06_run6; mkdir -p $out
export TMPDIR=/tmp
line() {
  python -c "
import json,sys,os
j=json.loads(sys.stdin.read()); r=j['roofline']; c=j['config']
print(f\"{os.environ.get('VNAME','shipped'):14s} {c['resolution']:5d}^2 x {c['cascades_per_gpu']:2d} {'fp16' if 'fp16' in c['workload'] else 'fp32'} group {c['cascades_per_launch']:2d}  {j['value']:9.0f} grids/s  step {j['ms_per_step']*1e3:8.1f} us  row {r['rowpass']['ms']*1e3:7.1f} us  col {r['colpass']['ms']*1e3:7.1f} us  step_frac {r['step_frac']:.3f}  on bytes moved {r['frac_of_peak_on_bytes_moved']['step']:.3f}\")"
}
run() { python bench.py --cpu-seconds 0 --no-frame --no-regime "$@" 2>/dev/null | line; }
use() { if [ "$1" = shipped ]; then unset DATUM_OCEAN_HIP_LIB; export VNAME=shipped; else export DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_$1.so); export VNAME=$1; fi; }
{
for rep in 1 2 3; do
  for v in shipped nofence; do use $v
    run --resolution 2048 --cascades 1 --steps 500 --warmup 50
    run --resolution 2048 --cascades 4 --steps 200 --warmup 20
    run --resolution 2048 --cascades 1 --steps 500 --warmup 50 --spectrum fp16
  done
done
unset DATUM_OCEAN_HIP_LIB
} > $out/col2048_fence.txt 2>&1
cat $out/col2048_fence.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; tail -5 $out/pytest_gpu.txt
cp gpurun_out/parity_table.txt $out/parity_table.txt 2>/dev/null
