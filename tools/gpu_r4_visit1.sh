#!/bin/bash
# round 4, one GPU visit: NTT variants (shift twiddles + cheap division vs the round-3 kernel), the squaring experiment, cold setup
mkdir -p gpurun_out/r4v1; export TMPDIR=/tmp
O=gpurun_out/r4v1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stark_large.py tests/test_gpu_stark_steps.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
OLD=eigen-zkvm_amd/variants/libzkgpu_nttold.so; SQR=eigen-zkvm_amd/variants/libzkgpu_sqr3.so
for rep in 1 2; do
  echo "== new" >> $O/ntt.txt; timeout 120 python tools/ntt_time.py 24 1 >> $O/ntt.txt 2>&1
  echo "== old" >> $O/ntt.txt; ZKGPU_LIB=$OLD timeout 120 python tools/ntt_time.py 24 1 >> $O/ntt.txt 2>&1
done
for cfg in "22 1" "20 8" "18 36"; do
  echo "== new $cfg" >> $O/ntt.txt; timeout 120 python tools/ntt_time.py $cfg >> $O/ntt.txt 2>&1
  echo "== old $cfg" >> $O/ntt.txt; ZKGPU_LIB=$OLD timeout 120 python tools/ntt_time.py $cfg >> $O/ntt.txt 2>&1
done
echo "== new" >> $O/lde.txt; timeout 200 python tools/lde_time.py 24 19 36 >> $O/lde.txt 2>&1
echo "== old" >> $O/lde.txt; ZKGPU_LIB=$OLD timeout 200 python tools/lde_time.py 24 19 36 >> $O/lde.txt 2>&1
echo "== new" >> $O/lde.txt; timeout 200 python tools/lde_time.py 20 19 36 >> $O/lde.txt 2>&1
echo "== old" >> $O/lde.txt; ZKGPU_LIB=$OLD timeout 200 python tools/lde_time.py 20 19 36 >> $O/lde.txt 2>&1
cat $O/ntt.txt $O/lde.txt | grep -v amdgpu.ids
for rep in 1 2; do
  echo "== shipped" >> $O/poseidon.txt; timeout 120 python tools/merkle_bench.py 22 19 22 36 >> $O/poseidon.txt 2>&1
  echo "== sqr3" >> $O/poseidon.txt; ZKGPU_LIB=$SQR timeout 120 python tools/merkle_bench.py 22 19 22 36 >> $O/poseidon.txt 2>&1
done
grep -v amdgpu.ids $O/poseidon.txt
bash tools/gpu_pmc_sq.sh > $O/pmc_sq_shipped.txt 2>&1; grep "^zk::" $O/pmc_sq_shipped.txt | head -12
ZKGPU_LIB=$SQR bash tools/gpu_pmc_sq.sh > $O/pmc_sq_sqr3.txt 2>&1; grep "^zk::" $O/pmc_sq_sqr3.txt | head -12
timeout 300 python tools/cold_setup_time.py 20 2>&1 | grep -v amdgpu.ids | tee $O/cold_setup.txt
timeout 300 python tools/cold_setup_time.py 24 2>&1 | grep -v amdgpu.ids | tee -a $O/cold_setup.txt
