S="--standin-peers 7 --payload xyz32 --standin-gbps 300 --standin-workgroups 32"
run() { python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-frame --no-regime "$@" 2>/dev/null | python3 -c '
import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        j=json.loads(l); r=j["roofline"]; print("  %-60s %8.0f grids/s  row %6.2f us col %6.2f us  compute %.3f ms gather %.3f ms" % (sys.argv[1], j["value"], r["rowpass"]["ms"]*1e3, r["colpass"]["ms"]*1e3, j["compute_ms"], j["gather_ms"]))' "$LABEL"; }
for rep in 1 2; do
LABEL="alone" run
LABEL="alone on 224 CUs" DATUM_COMPUTE_CUMASK=$(python3 -c "print(hex(((1<<256)-1) ^ 0xFFFFFFFF))") run
for m in 0 7 9 1; do
LABEL="comm_cus 32, mode $m" DATUM_STANDIN_MODE=$m run $S --comm-cus 32
LABEL="comm_cus 0, mode $m" DATUM_STANDIN_MODE=$m run $S --comm-cus 0
done
done
