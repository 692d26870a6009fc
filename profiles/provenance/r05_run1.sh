#!/bin/bash
# Run ON the GPU box (from the repository root): round 5's per-exchange LDS layouts -- parity suite, then the step kernels against round 4's build
# (datum_amd/lib/libdatum_ocean_hip_r04.so = the module as round 4 shipped it), interleaved, at every BASELINE size
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05/run1_tests.txt
cat gpurun_out/r05/run1_tests.txt
for rep in 1 2; do
  for lib in shipped datum_amd/lib/libdatum_ocean_hip_r04.so; do
    echo "== $lib (pass $rep)"
    if [ "$lib" = shipped ]; then tools/sizes.sh; else tools/sizes.sh $lib; fi
  done
done 2>&1 | tee gpurun_out/r05/run1_sizes.txt
