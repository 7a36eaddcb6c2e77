#!/bin/bash
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_bn128.py tests/test_gpu_stark_prove.py -x -q -m gpu 2>&1 | tail -3
python3 tools/final_stark_probe.py 5 2>&1 | tail -1
