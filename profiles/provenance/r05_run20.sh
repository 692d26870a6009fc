S="--standin-peers 7 --payload xyz32 --standin-gbps 300 --standin-workgroups 32"
run() { python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-frame --no-regime "$@" 2>&1 | python3 -c '
import json,sys
ok=False
for l in sys.stdin:
    if l.startswith("{"):
        ok=True
        j=json.loads(l); r=j["roofline"]; print("  %-60s %8.0f grids/s  row %6.2f us col %6.2f us  compute %.3f ms gather %.3f ms" % (sys.argv[1], j["value"], r["rowpass"]["ms"]*1e3, r["colpass"]["ms"]*1e3, j["compute_ms"], j["gather_ms"]))
    elif "Error" in l or "error" in l: print(l.rstrip())
if not ok: print("  %s: no result" % sys.argv[1])' "$LABEL"; }
for rep in 1 2; do
for cc in 32 0; do
LABEL="comm_cus $cc, gathered: torch allocator" run $S --comm-cus $cc
LABEL="comm_cus $cc, gathered: fine-grained" DATUM_GATHERED_FLAGS=1 run $S --comm-cus $cc
LABEL="comm_cus $cc, gathered: uncached" DATUM_GATHERED_FLAGS=3 run $S --comm-cus $cc
LABEL="comm_cus $cc, gathered: torch, nt stores (mode 4)" DATUM_STANDIN_MODE=4 run $S --comm-cus $cc
LABEL="comm_cus $cc, gathered: uncached, nt (mode 4)" DATUM_STANDIN_MODE=4 DATUM_GATHERED_FLAGS=3 run $S --comm-cus $cc
done
done
