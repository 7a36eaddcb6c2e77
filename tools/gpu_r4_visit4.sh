#!/bin/bash
# round 4, GPU visit 4: cold setup with one compiler process per step program
mkdir -p gpurun_out/r4v4; export TMPDIR=/tmp
O=gpurun_out/r4v4
export AMD_COMGR_CACHE=0
for nb in 20 24; do
  echo "== one after the other, in process (ZK_JIT_SERIAL=1)" >> $O/cold_setup.txt; ZK_JIT_SERIAL=1 timeout 300 python tools/cold_setup_time.py $nb 2>&1 | grep -v amdgpu.ids >> $O/cold_setup.txt
  echo "== three host threads, in process (ZK_JIT_INPROCESS=1)" >> $O/cold_setup.txt; ZK_JIT_INPROCESS=1 timeout 300 python tools/cold_setup_time.py $nb 2>&1 | grep -v amdgpu.ids >> $O/cold_setup.txt
  echo "== three host threads, a compiler process each (shipped)" >> $O/cold_setup.txt; timeout 300 python tools/cold_setup_time.py $nb 2>&1 | grep -v amdgpu.ids >> $O/cold_setup.txt
done
cat $O/cold_setup.txt
unset AMD_COMGR_CACHE
timeout 900 python -m pytest tests/test_gpu_stark_prove.py tests/test_gpu_round4.py tests/test_gpu_verify.py tests/test_gpu_stark_concurrent.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
