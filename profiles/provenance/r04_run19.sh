#!/bin/bash
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "handles_come_and_go or native_farm" 2>&1 | tail -15
