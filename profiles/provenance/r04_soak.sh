#!/bin/bash
# Run ON the GPU box: the GPU suite three times over (flakiness), then the freeze set
for i in 1 2 3; do python -m pytest tests -q -m gpu -x 2>&1 | tail -1; done
bash tools/r04_freeze.sh
