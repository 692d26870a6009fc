#!/bin/bash
# Run ON the GPU box: round-4 first pass -- GPU tests, the driver's bench line, the farm example, gather overhead with nt modes
mkdir -p gpurun_out/r04a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04a/tests.log
tail -5 gpurun_out/r04a/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a/bench20.json 2> gpurun_out/r04a/bench20.err; echo "bench exit $?"
timeout 300 python bench.py --steps 20 --warmup 5 --force-collective --cpu-seconds 0 --no-frame --no-regime > gpurun_out/r04a/bench20_farm1.json 2> gpurun_out/r04a/bench20_farm1.err; echo "bench farm exit $?"
timeout 300 ./examples/ocean_farm 1 2048 3 20 1 > gpurun_out/r04a/farm_example.txt 2>&1; echo "farm example exit $?"
timeout 900 bash tools/gather_overhead.sh > gpurun_out/r04a/gather_overhead.txt 2>&1; echo "gather overhead exit $?"
