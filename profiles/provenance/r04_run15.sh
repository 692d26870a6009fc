#!/bin/bash
# (belongs to commit 6d1cccf: the one-workgroup 64 x 64 step, removed since; DATUM_OCEAN_STEP64 no longer exists)
mkdir -p gpurun_out/r04n
python -m pytest tests -q -m gpu -x > gpurun_out/r04n/tests.log 2>&1; tail -3 gpurun_out/r04n/tests.log
frame() { python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-regime 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   reference frame n64 %.2f us (displace only %.2f), gen from 1024^2 maps %.2f us' % (j['reference_frame_n64']['us_per_frame'], j['reference_frame_n64']['us_displace_only'], j['gen']['ms']*1e3))"; }
{
for rep in 1 2 3; do
echo "one workgroup per cascade (shipped):"; frame
echo "two kernels (DATUM_OCEAN_STEP64=0):"; DATUM_OCEAN_STEP64=0 frame
done
echo "== 64^2 x 1 / x 4 / x 16, 5000 steps, bench.py"
for c in 1 4 16; do
python bench.py --resolution 64 --cascades $c --steps 5000 --warmup 200 --cpu-seconds 0 --no-regime --no-frame 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('   one workgroup: x $c', round(j['value']), 'grids/s', round(j['ms_per_step']*1e3,2), 'us/step', r['kernel'], round(r['rowpass']['ms']*1e3,2), round(r['colpass']['ms']*1e3,2))"
DATUM_OCEAN_STEP64=0 python bench.py --resolution 64 --cascades $c --steps 5000 --warmup 200 --cpu-seconds 0 --no-regime --no-frame 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('   two kernels:   x $c', round(j['value']), 'grids/s', round(j['ms_per_step']*1e3,2), 'us/step', r['kernel'], round(r['rowpass']['ms']*1e3,2), round(r['colpass']['ms']*1e3,2))"
done
} > gpurun_out/r04n/step64.txt 2>&1
cat gpurun_out/r04n/step64.txt
