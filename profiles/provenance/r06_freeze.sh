#!/bin/bash
# Run ON the GPU box: the round's freeze -- the GPU suite (verbose log), smoke, the driver's line (default and 20 steps after 5), sizes, and the profile set (kernel trace + PMC passes per size)
out=gpurun_out/r06_freeze; mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err; cut -c1-200 $out/bench_default.json
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 2>/dev/null; done > $out/bench_20steps.json; python3 -c "
import json
for l in open('$out/bench_20steps.json'):
    j=json.loads(l); print('20 steps:', round(j['value']), 'grids/s frac', round(j['roofline']['frac'],3), j['roofline']['kernel'], 'regime', round(j['roofline']['hbm_regime']['grids_per_s']), round(j['roofline']['hbm_regime']['frac_on_bytes_moved'],3))"
line() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; c=j['config']
print(f\"{c['resolution']:5d}^2 x {c['cascades_per_gpu']:2d} {sys.argv[1]:7s} group {c['cascades_per_launch']:2d} maps {c['map_stores'][:8]:8s} {j['value']:9.0f} grids/s  step {j['ms_per_step']*1e3:8.1f} us  row {r['rowpass']['ms']*1e3:7.1f} us  col {r['colpass']['ms']*1e3:7.1f} us  step_frac(survey bytes) {r['step_frac']:.3f}  on bytes moved {r['frac_of_peak_on_bytes_moved']['step']:.3f}\")" $1; }
run() { spec=$1; shift; python bench.py --cpu-seconds 0 --no-frame --no-regime --spectrum $spec "$@" 2>/dev/null | line $spec; }
{
for rep in 1 2; do
run fp32 --resolution 512 --cascades 1 --steps 2000 --warmup 200
run fp32 --resolution 512 --cascades 4 --steps 2000 --warmup 200
run fp32 --resolution 1024 --cascades 4 --steps 2000 --warmup 100
run fp32 --resolution 1024 --cascades 8 --steps 500 --warmup 50
run fp32 --resolution 1024 --cascades 16 --steps 200 --warmup 20
run fp32 --resolution 2048 --cascades 1 --steps 500 --warmup 50
run fp32 --resolution 2048 --cascades 4 --steps 200 --warmup 20
run fp32 --resolution 4096 --cascades 1 --steps 200 --warmup 20
run fp16 --resolution 4096 --cascades 1 --steps 200 --warmup 20
run fp16h0 --resolution 4096 --cascades 1 --steps 200 --warmup 20
run fp16h0 --resolution 1024 --cascades 4 --steps 2000 --warmup 100
done
} > $out/sizes.txt 2>&1; cat $out/sizes.txt
R=r06 tools/profile_all.sh
for s in 1024x4 512x1 2048x1 2048x4 4096 4096h 4096h0 1024x16; do cp gpurun_out/r06_prof_$s/summary.txt $out/summary_$s.txt 2>/dev/null; done
cp gpurun_out/r06_prof_1024x4/trace/*/*kernel_stats.csv $out/kernel_stats_1024x4.csv 2>/dev/null
cp gpurun_out/r06_traffic.json gpurun_out/r06_gpu_tests.txt $out/ 2>/dev/null
# the driver's line once more, now that profiles/ could hold this build's traffic file (it does not travel back by itself: copied below for the record)
cp gpurun_out/r06_traffic.json profiles/r06_traffic.json; python bench.py > $out/bench_default_with_traffic.json 2>/dev/null; python3 -c "
import json; j=json.loads(open('$out/bench_default_with_traffic.json').read()); r=j['roofline']; print('with traffic:', round(j['value']), r['frac'], r['traffic'], r['traffic_source'])"
