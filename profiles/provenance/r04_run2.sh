#!/bin/bash
mkdir -p gpurun_out/r04b
nproc > gpurun_out/r04b/cpus.txt; taskset -p $$ >> gpurun_out/r04b/cpus.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/r04b/cpus.txt 2>&1; python -c "import os;print(len(os.sched_getaffinity(0)), os.cpu_count())" >> gpurun_out/r04b/cpus.txt
timeout 600 python tools/dbg/two_oceans_trace.py 4400 > gpurun_out/r04b/trace.txt 2>&1; echo "trace exit $?"
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04b/tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04b/tests.log
tail -15 gpurun_out/r04b/tests.log
