#!/bin/bash
# Run ON the GPU box (round 6, second call): the row-pass skeleton stream by stream, cascade groups with the work spectrum reused from group to group
out=gpurun_out/r06_run2; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cascade_groups or rowpass_stage or rowpass_pins or four_cascades or sim_stage" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
SKEL_STREAMS=1 timeout 300 ./tools/dbg/bin/skeleton4096 > $out/skeleton4096_streams.txt 2>&1; cat $out/skeleton4096_streams.txt

line() {
  python -c "
import json,sys,os
j=json.loads(sys.stdin.read()); r=j['roofline']; c=j['config']
print(f\"{os.environ.get('VNAME','shipped'):14s} {c['resolution']:5d}^2 x {c['cascades_per_gpu']:2d} {'fp16' if 'fp16' in c['workload'] else 'fp32'} group {c['cascades_per_launch']:2d}  {j['value']:9.0f} grids/s  step {j['ms_per_step']*1e3:8.1f} us  row {r['rowpass']['ms']*1e3:7.1f} us  col {r['colpass']['ms']*1e3:7.1f} us  step_frac {r['step_frac']:.3f}  on bytes moved {r['frac_of_peak_on_bytes_moved']['step']:.3f}\")"
}
run() { python bench.py --cpu-seconds 0 --no-frame --no-regime "$@" 2>/dev/null | line; }
{
for rep in 1 2; do
  for g in 8 0 2; do run --resolution 1024 --cascades 8 --steps 300 --warmup 30 --cascade-group $g; done
  for g in 16 0 8 2; do run --resolution 1024 --cascades 16 --steps 200 --warmup 20 --cascade-group $g; done
  for g in 4 0 2; do run --resolution 2048 --cascades 4 --steps 200 --warmup 20 --cascade-group $g; done
  for g in 2 0; do run --resolution 2048 --cascades 2 --steps 200 --warmup 20 --cascade-group $g; done
  for g in 6 0 2; do run --resolution 1024 --cascades 6 --steps 300 --warmup 30 --cascade-group $g --spectrum fp16; done
  for g in 4 0 1; do run --resolution 512 --cascades 4 --steps 1000 --warmup 100 --cascade-group $g; done
done
} > $out/cascade_groups.txt 2>&1
cat $out/cascade_groups.txt
