#!/bin/bash
# Run ON the GPU box (one GPU): what a busy second stream costs the displacement step.  The all-gather of N ranks cannot run
# here; in its place (bench.py --standin-peers 7) a kernel of W workgroups stays resident on the communication stream for as
# long as a collective at G GB/s of bus bandwidth would and writes 7 peers' payloads into the gathered buffer.
STEPS=${STEPS:-20}; WARM=${WARM:-5}
run() { python bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --cpu-seconds 0 --no-frame "$@" 2>/dev/null | python3 -c '
import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        j=json.loads(l); print("  %-86s %8.0f grids/s  %7.2f us/step  compute %.3f ms gather %.3f ms" % (sys.argv[1], j["value"], j["ms_per_step"]*1e3, j["compute_ms"], j["gather_ms"]))
' "$*"; }
for rep in 1 2; do
echo "== repeat $rep, $STEPS steps after $WARM warm-up, 1024^2 x 4"
run
run --standin-peers 7 --payload xyz32 --standin-workgroups 0
run --standin-peers 7 --payload xyz32 --standin-workgroups 32 --standin-gbps 0
run --standin-peers 7 --payload xyz32 --standin-workgroups 32 --standin-gbps 300
run --standin-peers 7 --payload xyz32 --standin-workgroups 16 --standin-gbps 300
run --standin-peers 7 --payload xyz32 --standin-workgroups 64 --standin-gbps 300
run --standin-peers 7 --payload xyz16 --standin-workgroups 32 --standin-gbps 300
run --standin-peers 7 --payload maps --standin-workgroups 32 --standin-gbps 300
run --standin-peers 7 --payload xyz32 --standin-workgroups 32 --standin-gbps 300 --gather-every 1
run --standin-peers 7 --payload xyz32 --standin-workgroups 32 --standin-gbps 300 --gather-every 4
run --force-collective --payload xyz32
run --force-collective --payload xyz32 --gather-every 1
done
echo "== which part of the stand-in costs the step (32 workgroups, paced to 300 GB/s = 1.18 ms resident, xyz32)"
for m in 0 1 2 3; do echo "  mode $m (0 copy, 1 resident only, 2 reads only, 3 writes only):"; DATUM_STANDIN_MODE=$m run --standin-peers 7 --payload xyz32 --standin-workgroups 32 --standin-gbps 300; done
echo "== the same with non-temporal loads / stores in the stand-in (cache eviction or bandwidth contention?)"
for m in 4 5 6; do echo "  mode $m (4 copy nt, 5 reads only nt, 6 writes only nt):"; DATUM_STANDIN_MODE=$m run --standin-peers 7 --payload xyz32 --standin-workgroups 32 --standin-gbps 300; done
for m in 0 4; do echo "  mode $m unpaced (as fast as HBM takes it):"; DATUM_STANDIN_MODE=$m run --standin-peers 7 --payload xyz32 --standin-workgroups 32 --standin-gbps 0; done
echo "  mode 1 with 256 workgroups:"; DATUM_STANDIN_MODE=1 run --standin-peers 7 --payload xyz32 --standin-workgroups 256 --standin-gbps 300
echo "  200-step batches:"; STEPS=200 WARM=20 run; STEPS=200 WARM=20 run --standin-peers 7 --payload xyz32 --standin-workgroups 32 --standin-gbps 300
echo "== configs[3] share: 2048^2 x 1"
run --resolution 2048 --cascades 1
run --resolution 2048 --cascades 1 --standin-peers 7 --standin-workgroups 32 --standin-gbps 300
run --resolution 2048 --cascades 1 --standin-peers 7 --standin-workgroups 32 --standin-gbps 300 --gather-every 1
echo "== python bench.py --gpus 2 on a one-GPU box (own launcher; must fail cleanly, not hang)"
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-seconds 0 --no-frame > /tmp/two.out 2> /tmp/two.err; echo "  exit code $?; stdout lines: $(wc -l < /tmp/two.out); last stderr: $(tail -1 /tmp/two.err)"
