#!/bin/bash
# Run ON the GPU box (one GPU): round 5's questions about the all-gather stand-in.  The stand-in is a kernel of 32 workgroups on a
# second stream that stays resident for as long as a collective at 300 GB/s of bus bandwidth would (1.18 ms for xyz32 at 8 ranks), reads this rank's
# payload once per peer and writes 7 peers' payloads (bench.py --standin-peers 7; datum_amd/csrc/farm_standin.hip).  20-step batches, 1024^2 x 4.
#   modes: 0 copy, 1 resident only, 2 reads only, 3 writes only, 7 copy with the destination wrapped into one peer's share (same bytes, 1/7 of the
#   footprint), 8 writes only, wrapped
STEPS=${STEPS:-20}; WARM=${WARM:-5}
run() { python bench.py --steps $STEPS --warmup $WARM --cpu-seconds 0 --no-frame --no-regime --comm-cus 0 "$@" 2>/dev/null | python3 -c '
import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        j=json.loads(l); r=j["roofline"]; print("  %-64s %8.0f grids/s  %7.2f us/step  row %6.2f us col %6.2f us  compute %.3f ms gather %.3f ms" % (sys.argv[1], j["value"], j["ms_per_step"]*1e3, r["rowpass"]["ms"]*1e3, r["colpass"]["ms"]*1e3, j["compute_ms"], j["gather_ms"]))
' "$LABEL"; }
S="--standin-peers 7 --payload xyz32 --standin-workgroups 32 --standin-gbps 300"
for rep in 1 2 3; do
echo "== repeat $rep"
LABEL="no second stream" run
for m in 0 1 2 3 7 8; do LABEL="stand-in mode $m" DATUM_STANDIN_MODE=$m run $S; done
LABEL="mode 0, compute stream at high priority" DATUM_COMPUTE_PRIORITY=-1 run $S
LABEL="mode 0, 4 slices, one every 5 steps" DATUM_STANDIN_CHUNKS=4 run $S
LABEL="mode 0, 4 slices + compute stream at high priority" DATUM_STANDIN_CHUNKS=4 DATUM_COMPUTE_PRIORITY=-1 run $S
LABEL="mode 0, xyz16 payload" run --standin-peers 7 --payload xyz16 --standin-workgroups 32 --standin-gbps 300
LABEL="mode 0, 64 workgroups" run --standin-peers 7 --payload xyz32 --standin-workgroups 64 --standin-gbps 300
LABEL="mode 0, 16 workgroups" run --standin-peers 7 --payload xyz32 --standin-workgroups 16 --standin-gbps 300
done
echo "== 1024^2 x 16 (the step's working set is beyond the Infinity Cache with or without the gathered buffer), 10 steps"
STEPS=10 WARM=3
LABEL="x16: no second stream" run --cascades 16
LABEL="x16: stand-in mode 0" run --cascades 16 $S
LABEL="x16: stand-in mode 7 (wrapped)" DATUM_STANDIN_MODE=7 run --cascades 16 $S
echo "== 2048^2 x 1 (configs[3]'s tile), 20 steps"
STEPS=20 WARM=5
LABEL="2048: no second stream" run --resolution 2048 --cascades 1
LABEL="2048: stand-in mode 0" run --resolution 2048 --cascades 1 $S
LABEL="2048: stand-in mode 7 (wrapped)" DATUM_STANDIN_MODE=7 run --resolution 2048 --cascades 1 $S
LABEL="2048: mode 0 + compute stream at high priority" DATUM_COMPUTE_PRIORITY=-1 run --resolution 2048 --cascades 1 $S
python - <<'PY'
import torch
print("stream priority range (least, greatest):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
PY
