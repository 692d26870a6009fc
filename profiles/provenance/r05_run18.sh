S="--standin-peers 7 --payload xyz32 --standin-gbps 300"
run() { python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-frame --no-regime "$@" 2>/dev/null | python3 -c '
import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        j=json.loads(l); r=j["roofline"]; print("  %-60s %8.0f grids/s  row %6.2f us col %6.2f us  compute %.3f ms gather %.3f ms" % (sys.argv[1], j["value"], r["rowpass"]["ms"]*1e3, r["colpass"]["ms"]*1e3, j["compute_ms"], j["gather_ms"]))' "$LABEL"; }
for rep in 1 2; do
echo "== 1024^2 x 4, repeat $rep"
LABEL="alone" run
for cc in 0 16 24 32 40 48 64; do LABEL="stand-in 32 workgroups, comm_cus $cc" run $S --standin-workgroups 32 --comm-cus $cc; done
for cc in 32 64; do LABEL="stand-in 64 workgroups, comm_cus $cc" run $S --standin-workgroups 64 --comm-cus $cc; done
LABEL="stand-in 32 workgroups, comm_cus 32, 4 slices" DATUM_STANDIN_CHUNKS=4 run $S --standin-workgroups 32 --comm-cus 32
done
echo "== 2048^2 x 1"
LABEL="alone" run --resolution 2048 --cascades 1
for cc in 0 24 32 48 64; do LABEL="stand-in 32 workgroups, comm_cus $cc" run --resolution 2048 --cascades 1 $S --standin-workgroups 32 --comm-cus $cc; done
echo "== 1024^2 x 4, fp16 payload"
for cc in 0 32; do LABEL="xyz16, comm_cus $cc" run --standin-peers 7 --payload xyz16 --standin-gbps 300 --standin-workgroups 32 --comm-cus $cc; done
