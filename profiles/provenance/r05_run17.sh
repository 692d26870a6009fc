timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_shim.py tests/test_gpu_bench.py -m gpu -x -q -k "farm or gather or example or bench or collective" 2>&1 | tail -3
./examples/ocean_farm 1 1024 3 20 1 32 | tail -3
./examples/ocean_farm 1 1024 3 20 1 0 | tail -1
S="--standin-peers 7 --payload xyz32 --standin-workgroups 32 --standin-gbps 300"
for rep in 1 2; do
for cc in 0 32; do
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-frame --no-regime --comm-cus $cc $S 2>/dev/null | python3 -c '
import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        j=json.loads(l); print(j["value"], j["ms_per_step"], j["compute_ms"], j["gather_ms"], j["config"]["cu_partition"])'
done
done
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-frame --no-regime --force-collective 2>&1 | tail -1 | cut -c1-400
