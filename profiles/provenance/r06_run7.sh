#!/bin/bash
# Run ON the GPU box (round 6, seventh call): the GPU suite, the driver's bench line, and the round's profile set (kernel trace + PMC passes per size)
out=gpurun_out/r06_run7; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err; cut -c1-600 $out/bench_default.json
R=r06 tools/profile_all.sh
for s in 1024x4 512x1 2048x1 2048x4 4096 4096h; do cp gpurun_out/r06_prof_$s/summary.txt $out/summary_$s.txt 2>/dev/null; done
