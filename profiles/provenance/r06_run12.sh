#!/bin/bash
# Run ON the GPU box (round 6, twelfth call): the whole GPU suite on the build with the map store policy (ABI 8), the three fuzzers, the stand-in through the new policy
out=gpurun_out/r06_run12; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; grep -E "passed|failed" $out/pytest_gpu.txt
for f in "fuzz.py 60 7100" "api_fuzz.py 40 7200 40" "host_fuzz.py 30 7300 80"; do timeout 900 python tools/dbg/$f > $out/fuzz_$(echo $f | cut -d. -f1).txt 2>&1; tail -1 $out/fuzz_$(echo $f | cut -d. -f1).txt; done
for pol in "written through" auto; do
  python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-frame --no-regime --standin-peers 7 --payload xyz32 --standin-workgroups 32 --standin-gbps 300 --map-stores "$pol" 2>/dev/null | python3 -c '
import json,sys
j=json.loads(sys.stdin.read()); print(j["config"]["map_stores"], "|", round(j["value"]), "grids/s  compute", round(j["compute_ms"],3), "ms  gather", round(j["gather_ms"],3), "ms", j["config"]["cu_partition"])'
done
