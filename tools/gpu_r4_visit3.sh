#!/bin/bash
# round 4, GPU visit 3: cold setup after the power kernel calls its products; 2^24 golden test; whole suite
mkdir -p gpurun_out/r4v3; export TMPDIR=/tmp
O=gpurun_out/r4v3
export AMD_COMGR_CACHE=0
for nb in 20 24; do
  echo "== serial" >> $O/cold_setup.txt; ZK_JIT_SERIAL=1 timeout 300 python tools/cold_setup_time.py $nb 2>&1 | grep -v amdgpu.ids >> $O/cold_setup.txt
  echo "== concurrent" >> $O/cold_setup.txt; timeout 300 python tools/cold_setup_time.py $nb 2>&1 | grep -v amdgpu.ids >> $O/cold_setup.txt
done
cat $O/cold_setup.txt
unset AMD_COMGR_CACHE
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; tail -12 $O/pytest.log
