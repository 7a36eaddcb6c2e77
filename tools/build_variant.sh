#!/bin/bash
# tools/build_variant.sh NAME SRC.hip "extra flags": an experimental libzkgpu with one translation unit rebuilt under other flags
# -> eigen-zkvm_amd/variants/libzkgpu_NAME.so (git-ignored; select with ZKGPU_LIB=...)
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/../eigen-zkvm_amd/csrc"
mkdir -p ../variants /tmp/zkvar
obj=/tmp/zkvar/${name}_${src%.hip}.o
extra=""; [ "$src" = msm.hip ] && extra="--gpu-max-threads-per-block=64"; { [ "$src" = poseidon.hip ] || [ "$src" = ntt.hip ] || [ "$src" = frhash.hip ] || [ "$src" = frhash_bls12381.hip ]; } && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed $extra $flags -c $src -o $obj
objs=""; for o in ntt poseidon frhash frhash_bls12381 stark expr_jit msm stark_prover stark_verify starkinfo_gen groth16 compressor12 capi; do
  if [ "$o.hip" = "$src" ]; then objs="$objs $obj"; else objs="$objs $o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libzkgpu_${name}.so $objs -L/opt/rocm/lib -lhiprtc
echo built ../variants/libzkgpu_${name}.so
