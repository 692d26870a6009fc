#!/bin/bash
# Run ON the GPU box (round 6, fifth call): the streamed-maps threshold (x 4 / 5 / 6 of 1024^2 written through against streamed), cascade groups under the new policy,
# the inputs-first prologue once more
out=gpurun_out/r06_run5; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cascade_groups or golden_n64 or end_to_end or fp16" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
line() {
  python -c "
import json,sys,os
j=json.loads(sys.stdin.read()); r=j['roofline']; c=j['config']
print(f\"{os.environ.get('VNAME','shipped'):14s} {c['resolution']:5d}^2 x {c['cascades_per_gpu']:2d} {'fp16' if 'fp16' in c['workload'] else 'fp32'} group {c['cascades_per_launch']:2d}  {j['value']:9.0f} grids/s  step {j['ms_per_step']*1e3:8.1f} us  row {r['rowpass']['ms']*1e3:7.1f} us  col {r['colpass']['ms']*1e3:7.1f} us  step_frac {r['step_frac']:.3f}  on bytes moved {r['frac_of_peak_on_bytes_moved']['step']:.3f}\")"
}
run() { python bench.py --cpu-seconds 0 --no-frame --no-regime "$@" 2>/dev/null | line; }
use() { if [ "$1" = shipped ]; then unset DATUM_OCEAN_HIP_LIB; export VNAME=shipped; else export DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_$1.so); export VNAME=$1; fi; }
{
for rep in 1 2; do
  for v in never_stream always_stream; do use $v
    run --resolution 1024 --cascades 4 --steps 1000 --warmup 100
    run --resolution 1024 --cascades 5 --steps 500 --warmup 50
    run --resolution 1024 --cascades 6 --steps 500 --warmup 50
    run --resolution 2048 --cascades 1 --steps 300 --warmup 30
    run --resolution 1024 --cascades 4 --steps 1000 --warmup 100 --spectrum fp16
    run --resolution 1024 --cascades 6 --steps 500 --warmup 50 --spectrum fp16
    run --resolution 512 --cascades 16 --steps 500 --warmup 50
  done
  use shipped
  for g in 8 4 2; do run --resolution 1024 --cascades 8 --steps 300 --warmup 30 --cascade-group $g; done
  for g in 16 8 4; do run --resolution 1024 --cascades 16 --steps 200 --warmup 20 --cascade-group $g; done
  for g in 12 6 4 3; do run --resolution 1024 --cascades 12 --steps 200 --warmup 20 --cascade-group $g; done
  for g in 4 2 1; do run --resolution 2048 --cascades 4 --steps 200 --warmup 20 --cascade-group $g; done
  for g in 2 1; do run --resolution 2048 --cascades 2 --steps 200 --warmup 20 --cascade-group $g; done
  for g in 8 4; do run --resolution 1024 --cascades 8 --steps 300 --warmup 30 --cascade-group $g --spectrum fp16; done
  for g in 2 1; do run --resolution 2048 --cascades 2 --steps 200 --warmup 20 --cascade-group $g --spectrum fp16; done
  for v in shipped inputs_first; do use $v
    run --resolution 4096 --cascades 1 --steps 100 --warmup 10 --spectrum fp16
    run --resolution 4096 --cascades 1 --steps 100 --warmup 10
    run --resolution 1024 --cascades 4 --steps 2000 --warmup 100
    run --resolution 1024 --cascades 8 --steps 300 --warmup 30
  done
done
unset DATUM_OCEAN_HIP_LIB
} > $out/policy_and_groups.txt 2>&1
cat $out/policy_and_groups.txt
