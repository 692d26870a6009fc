#!/bin/bash
# Run ON the GPU box (round 6, eighth call): h0 read as halves (DATUM_OCEAN_SPECTRUM_FP16_H0) -- its tests, then fp16 against fp16h0 at the sizes of the configs
out=gpurun_out/r06_run8; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_shim.py -x -q -m gpu -k "fp16 or halves or random_parameters or power_of_two or handles_come" > $out/pytest_h0h.txt 2>&1; tail -5 $out/pytest_h0h.txt
line() {
  python -c "
import json,sys,os
j=json.loads(sys.stdin.read()); r=j['roofline']; c=j['config']
print(f\"{c['resolution']:5d}^2 x {c['cascades_per_gpu']:2d} {os.environ['SPEC']:7s} group {c['cascades_per_launch']:2d}  {j['value']:9.0f} grids/s  step {j['ms_per_step']*1e3:8.1f} us  row {r['rowpass']['ms']*1e3:7.1f} us  col {r['colpass']['ms']*1e3:7.1f} us  step_frac(survey bytes) {r['step_frac']:.3f}  on bytes moved {r['frac_of_peak_on_bytes_moved']['step']:.3f}  row alone on bytes moved {r['frac_of_peak_on_bytes_moved']['rowpass']:.3f}\")"
}
run() { python bench.py --cpu-seconds 0 --no-frame --no-regime --spectrum $SPEC "$@" 2>/dev/null | line; }
{
for rep in 1 2 3; do
  for SPEC in fp16 fp16h0; do export SPEC
    run --resolution 4096 --cascades 1 --steps 200 --warmup 20
    run --resolution 2048 --cascades 1 --steps 500 --warmup 50
    run --resolution 2048 --cascades 4 --steps 200 --warmup 20
    run --resolution 1024 --cascades 4 --steps 2000 --warmup 100
    run --resolution 1024 --cascades 16 --steps 200 --warmup 20
  done
done
} > $out/h0_halves.txt 2>&1
cat $out/h0_halves.txt
