#!/bin/bash
# Run ON the GPU box (round 6): the 2048^2 row pass with 8 points per thread (512-thread pairs, 2048 = 8 x 8 x 8 x 4, 72 registers, three or four workgroups per CU) against the
# shipped 16 points per thread (256-thread pairs, 124 registers, four per CU)
out=gpurun_out/r06_run19; mkdir -p $out
export TMPDIR=/tmp
line() {
  python -c "
import json,sys,os
j=json.loads(sys.stdin.read()); r=j['roofline']; c=j['config']
print(f\"{os.environ.get('VNAME','shipped'):12s} {c['resolution']:5d}^2 x {c['cascades_per_gpu']:2d} {os.environ['SPEC']:7s} {j['value']:9.0f} grids/s  step {j['ms_per_step']*1e3:8.2f} us  row {r['rowpass']['ms']*1e3:7.2f} us  col {r['colpass']['ms']*1e3:7.2f} us\")"
}
run() { python bench.py --cpu-seconds 0 --no-frame --no-regime --spectrum $SPEC "$@" 2>/dev/null | line; }
use() { if [ "$1" = shipped ]; then unset DATUM_OCEAN_HIP_LIB; export VNAME=shipped; else export DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_$1.so); export VNAME=$1; fi; }
{
for rep in 1 2 3; do
  for v in shipped row2048e8 row2048e8x4; do use $v
    for SPEC in fp32 fp16h0; do export SPEC
      run --resolution 2048 --cascades 1 --steps 500 --warmup 50
      run --resolution 2048 --cascades 4 --steps 200 --warmup 20
    done
  done
done
unset DATUM_OCEAN_HIP_LIB
} > $out/row2048.txt 2>&1
cat $out/row2048.txt
export DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_row2048e8.so)
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "2048" 2>&1 | grep -E "passed|failed"
