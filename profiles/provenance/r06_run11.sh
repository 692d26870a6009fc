#!/bin/bash
# Run ON the GPU box (round 6, eleventh call): the all-gather stand-in (32 workgroups paced to 300 GB/s of bus bandwidth on a second stream, 1 / 3 / 7
# peers' xyz32 payloads = 2 / 4 / 8 ranks) under 20-step batches at configs[3]'s own shape (2048^2 x 1) and at the headline's (1024^2 x 4), with the streams on the whole
# device (--comm-cus 0) and on disjoint compute units (--comm-cus 32); "stream0" = the maps streamed (nt) although the handle's own working set fits the Infinity Cache
# (the stand-in's gathered buffer competes for it), shipped = written through
out=gpurun_out/r06_run11; mkdir -p $out
export TMPDIR=/tmp
run() { python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-frame --no-regime "$@" 2>/dev/null | python3 -c '
import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        j=json.loads(l); r=j["roofline"]; print("  %-8s %-62s %8.0f grids/s  %7.2f us/step  row %6.2f us col %6.2f us  compute %.3f ms gather %.3f ms" % (sys.argv[2], sys.argv[1], j["value"], j["ms_per_step"]*1e3, r["rowpass"]["ms"]*1e3, r["colpass"]["ms"]*1e3, j["compute_ms"], j["gather_ms"]))
' "$LABEL" "$VNAME"; }
use() { if [ "$1" = shipped ]; then unset DATUM_OCEAN_HIP_LIB; export VNAME=shipped; else export DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_$1.so); export VNAME=$1; fi; }
{
for rep in 1 2 3; do
  echo "== repeat $rep"
  for v in shipped stream0; do use $v
    for shape in "--resolution 2048 --cascades 1" "--resolution 1024 --cascades 4"; do
      LABEL="$shape: no second stream" run $shape --comm-cus 0
      for peers in 1 3 7; do
        S="--standin-peers $peers --payload xyz32 --standin-workgroups 32 --standin-gbps 300"
        LABEL="$shape: stand-in x $peers peers, whole device" run $shape $S --comm-cus 0
        LABEL="$shape: stand-in x $peers peers, 32 CUs of its own" run $shape $S --comm-cus 32
      done
    done
  done
done
unset DATUM_OCEAN_HIP_LIB
} > $out/standin.txt 2>&1
cat $out/standin.txt
