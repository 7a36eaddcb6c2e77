#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/msm_affine_ubench.py > gpurun_out/q_msm_ubench.log 2>&1; cat gpurun_out/q_msm_ubench.log | tail -12
