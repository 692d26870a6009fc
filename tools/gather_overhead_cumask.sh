#!/bin/bash
# Run ON the GPU box (one GPU): the all-gather stand-in with the communication stream and the compute stream on DISJOINT compute units
# (hipExtStreamCreateWithCUMask).  profiles/r05_gather_overhead.txt found the stand-in's cost to be its bursts in the in-order memory queues of the
# CUs it shares with the step's workgroups; here it gets CUs of its own.  20-step batches, 1024^2 x 4, stand-in mode 0 (copy), 300 GB/s, xyz32.
STEPS=${STEPS:-20}; WARM=${WARM:-5}
run() { python bench.py --steps $STEPS --warmup $WARM --cpu-seconds 0 --no-frame --no-regime --comm-cus 0 "$@" 2>/dev/null | python3 -c '
import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        j=json.loads(l); r=j["roofline"]; print("  %-72s %8.0f grids/s  %7.2f us/step  row %6.2f us col %6.2f us  compute %.3f ms gather %.3f ms" % (sys.argv[1], j["value"], j["ms_per_step"]*1e3, r["rowpass"]["ms"]*1e3, r["colpass"]["ms"]*1e3, j["compute_ms"], j["gather_ms"]))
' "$LABEL"; }
ALL=$(python3 -c 'print(hex((1<<256)-1))')
comp() { python3 -c "print(hex(((1<<256)-1) ^ $1))"; }
WG=${WG:-32}
S="--standin-peers 7 --payload xyz32 --standin-workgroups $WG --standin-gbps 300"
python tools/cumask_probe.py 2>&1 | grep -v amdgpu
for rep in 1 2; do
echo "== repeat $rep"
LABEL="no second stream" run
LABEL="no second stream, compute on 248 CUs" DATUM_COMPUTE_CUMASK=$(comp 0xFF) run
LABEL="no second stream, compute on 240 CUs" DATUM_COMPUTE_CUMASK=$(comp 0xFFFF) run
LABEL="no second stream, compute on 224 CUs" DATUM_COMPUTE_CUMASK=$(comp 0xFFFFFFFF) run
LABEL="stand-in, streams unmasked" run $S
for m in 0xFF 0xFFFF 0xFFFFFFFF; do
  LABEL="stand-in on CUs $m, compute unmasked" DATUM_COMM_CUMASK=$m run $S
  LABEL="stand-in on CUs $m, compute on the others" DATUM_COMM_CUMASK=$m DATUM_COMPUTE_CUMASK=$(comp $m) run $S
done
LABEL="stand-in on CUs 0xFFFF (64 workgroups), compute on the others" DATUM_COMM_CUMASK=0xFFFF DATUM_COMPUTE_CUMASK=$(comp 0xFFFF) run --standin-peers 7 --payload xyz32 --standin-workgroups 64 --standin-gbps 300
LABEL="stand-in on CUs 0xFFFF, compute on the others, 4 slices" DATUM_STANDIN_CHUNKS=4 DATUM_COMM_CUMASK=0xFFFF DATUM_COMPUTE_CUMASK=$(comp 0xFFFF) run $S
done
echo "== 2048^2 x 1 (configs[3]'s tile), 20 steps"
LABEL="2048: no second stream" run --resolution 2048 --cascades 1
LABEL="2048: stand-in, streams unmasked" run --resolution 2048 --cascades 1 $S
LABEL="2048: stand-in on CUs 0xFFFF, compute on the others" DATUM_COMM_CUMASK=0xFFFF DATUM_COMPUTE_CUMASK=$(comp 0xFFFF) run --resolution 2048 --cascades 1 $S
LABEL="2048: stand-in on CUs 0xFFFFFFFF, compute on the others" DATUM_COMM_CUMASK=0xFFFFFFFF DATUM_COMPUTE_CUMASK=$(comp 0xFFFFFFFF) run --resolution 2048 --cascades 1 $S
