#!/bin/bash
# Run ON the GPU box (round 6, first call): the new tests, the 4096^2 row-pass skeleton, cascade groups on / off beyond the Infinity Cache, and the
# inputs-first prologue (variant library) against the shipped one, interleaved.  Output: gpurun_out/r06_run1/
out=gpurun_out/r06_run1; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cascade_groups or golden_n64 or partitioned or literal" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
timeout 300 ./tools/dbg/bin/skeleton4096 > $out/skeleton4096.txt 2>&1; tail -5 $out/skeleton4096.txt

line() {
  python -c "
import json,sys,os
j=json.loads(sys.stdin.read()); r=j['roofline']; c=j['config']
print(f\"{os.environ.get('VNAME','shipped'):14s} {c['resolution']:5d}^2 x {c['cascades_per_gpu']:2d} {'fp16' if 'fp16' in c['workload'] else 'fp32'} group {c['cascades_per_launch']:2d}  {j['value']:9.0f} grids/s  step {j['ms_per_step']*1e3:8.1f} us  row {r['rowpass']['ms']*1e3:7.1f} us  col {r['colpass']['ms']*1e3:7.1f} us  step_frac {r['step_frac']:.3f}  on bytes moved {r['frac_of_peak_on_bytes_moved']['step']:.3f}\")"
}
run() { python bench.py --cpu-seconds 0 --no-frame --no-regime "$@" 2>/dev/null | line; }

# cascade groups (task 2): every cascade in one launch per pass (group = cascades) against the module's own groups (0)
{
for rep in 1 2; do
  for g in 8 0 1 2; do run --resolution 1024 --cascades 8 --steps 300 --warmup 30 --cascade-group $g; done
  for g in 16 0 2 8; do run --resolution 1024 --cascades 16 --steps 200 --warmup 20 --cascade-group $g; done
  for g in 4 0 2; do run --resolution 2048 --cascades 4 --steps 200 --warmup 20 --cascade-group $g; done
  for g in 2 0; do run --resolution 2048 --cascades 2 --steps 200 --warmup 20 --cascade-group $g; done
  for g in 6 0; do run --resolution 1024 --cascades 6 --steps 300 --warmup 30 --cascade-group $g --spectrum fp16; done
done
} > $out/cascade_groups.txt 2>&1
cat $out/cascade_groups.txt

# inputs-first prologue (task 1, the cheap part): shipped against the variant, interleaved
{
for rep in 1 2 3; do
  for lib in shipped datum_amd/lib/variants/lib_inputs_first.so; do
    if [ "$lib" = shipped ]; then unset DATUM_OCEAN_HIP_LIB; export VNAME=shipped; else export DATUM_OCEAN_HIP_LIB=$(realpath $lib); export VNAME=inputs_first; fi
    run --resolution 1024 --cascades 4 --steps 2000 --warmup 100
    run --resolution 4096 --cascades 1 --steps 200 --warmup 20 --spectrum fp16
    run --resolution 4096 --cascades 1 --steps 200 --warmup 20
    run --resolution 2048 --cascades 1 --steps 500 --warmup 50
    run --resolution 512 --cascades 1 --steps 2000 --warmup 200
  done
done
unset DATUM_OCEAN_HIP_LIB
} > $out/inputs_first.txt 2>&1
cat $out/inputs_first.txt
