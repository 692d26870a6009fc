#!/bin/bash
# Run ON the GPU box (round 6): extended seeded differential runs on the frozen build (other seeds than the suite's and than tools/r06_run12.sh's)
out=gpurun_out/r06_run13; mkdir -p $out
export TMPDIR=/tmp
for f in "fuzz.py 500 8100" "api_fuzz.py 300 8200 60" "host_fuzz.py 250 8300 120"; do n=$(echo $f | cut -d. -f1); timeout 2400 python tools/dbg/$f > $out/fuzz_$n.txt 2>&1; tail -1 $out/fuzz_$n.txt; done
