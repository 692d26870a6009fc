for rep in 1 2 3 4; do for s in 2 4 8 100; do DATUM_BENCH_STRIDE=$s python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-frame 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('stride $s', round(j['value']), 'us/step', round(j['ms_per_step']*1e3,2), 'compute_ms', round(j['compute_ms'],3), 'n', j['roofline']['launches_timed'], 'frac', round(j['roofline']['frac'],3))"; done; done
