#!/bin/bash
# what does RCCL do with two ranks on one GPU?  (bounded by timeout)
DATUM_FARM_DEVICES=1 timeout 90 ./examples/ocean_farm 2 256 1 2 > gpurun_out/farm_2on1.txt 2>&1; echo "exit $?" >> gpurun_out/farm_2on1.txt
cat gpurun_out/farm_2on1.txt | tail -20
