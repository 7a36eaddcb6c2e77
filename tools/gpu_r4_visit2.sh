#!/bin/bash
# round 4, GPU visit 2: truly cold setups (no code-object cache of ours, none of comgr's): step programs compiled one after the other vs side by side
mkdir -p gpurun_out/r4v2; export TMPDIR=/tmp
O=gpurun_out/r4v2
export AMD_COMGR_CACHE=0
for nb in 20 24; do
  echo "== serial" >> $O/cold_setup.txt; ZK_JIT_SERIAL=1 timeout 300 python tools/cold_setup_time.py $nb 2>&1 | grep -v amdgpu.ids >> $O/cold_setup.txt
  echo "== concurrent" >> $O/cold_setup.txt; timeout 300 python tools/cold_setup_time.py $nb 2>&1 | grep -v amdgpu.ids >> $O/cold_setup.txt
done
cat $O/cold_setup.txt
unset AMD_COMGR_CACHE
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stark_prove.py tests/test_gpu_merkle_large.py tests/test_gpu_round4.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 120 python tools/merkle_bench.py 22 19 22 36 24 10 2>&1 | grep -v amdgpu.ids | tee $O/poseidon.txt
