# the CU partition at smaller world sizes: the stand-in with 1 / 3 / 7 peers' payloads (2 / 4 / 8 ranks), 300 GB/s, comm_cus 0 / 8 / 16 / 32
run() { python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-frame --no-regime "$@" 2>/dev/null | python3 -c '
import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        j=json.loads(l); r=j["roofline"]; print("  %-44s %8.0f grids/s  row %6.2f us col %6.2f us  compute %.3f ms gather %.3f ms" % (sys.argv[1], j["value"], r["rowpass"]["ms"]*1e3, r["colpass"]["ms"]*1e3, j["compute_ms"], j["gather_ms"]))' "$LABEL"; }
for rep in 1 2; do
LABEL="alone" run
for peers in 1 3 7; do
for cc in 0 8 16 32; do
LABEL="$peers peers, comm_cus $cc" run --standin-peers $peers --payload xyz32 --standin-gbps 300 --standin-workgroups 32 --comm-cus $cc
done
done
done
