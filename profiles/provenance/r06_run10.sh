#!/bin/bash
# Run ON the GPU box (round 6, tenth call): the fp16-stored work spectrum in blocks of 8 x 16 values (128-byte block rows: the row pass writes whole lines;
# variant sbc16) against the shipped 8 x 8 (64-byte block rows, the other half of a line written by the neighbouring pair's workgroup)
out=gpurun_out/r06_run10; mkdir -p $out
export TMPDIR=/tmp
line() {
  python -c "
import json,sys,os
j=json.loads(sys.stdin.read()); r=j['roofline']; c=j['config']
print(f\"{os.environ.get('VNAME','shipped'):8s} {c['resolution']:5d}^2 x {c['cascades_per_gpu']:2d} {os.environ['SPEC']:7s} group {c['cascades_per_launch']:2d}  {j['value']:9.0f} grids/s  step {j['ms_per_step']*1e3:8.1f} us  row {r['rowpass']['ms']*1e3:7.1f} us  col {r['colpass']['ms']*1e3:7.1f} us  step_frac(survey bytes) {r['step_frac']:.3f}  on bytes moved {r['frac_of_peak_on_bytes_moved']['step']:.3f}\")"
}
run() { python bench.py --cpu-seconds 0 --no-frame --no-regime --spectrum $SPEC "$@" 2>/dev/null | line; }
use() { if [ "$1" = shipped ]; then unset DATUM_OCEAN_HIP_LIB; export VNAME=shipped; else export DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_$1.so); export VNAME=$1; fi; }
{
for rep in 1 2 3; do
  for v in shipped sbc16; do use $v
    for SPEC in fp16h0 fp16; do export SPEC
      run --resolution 4096 --cascades 1 --steps 200 --warmup 20
      run --resolution 2048 --cascades 1 --steps 500 --warmup 50
      run --resolution 2048 --cascades 4 --steps 200 --warmup 20
      run --resolution 1024 --cascades 4 --steps 2000 --warmup 100
      run --resolution 1024 --cascades 16 --steps 200 --warmup 20
      run --resolution 512 --cascades 1 --steps 2000 --warmup 100
    done
  done
done
unset DATUM_OCEAN_HIP_LIB
} > $out/sbc16.txt 2>&1
cat $out/sbc16.txt
export DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_sbc16.so)
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fp16 or halves or random_parameters or power_of_two or handles_come or rowpass_stage" > $out/pytest_sbc16.txt 2>&1; grep -E "passed|failed" $out/pytest_sbc16.txt
