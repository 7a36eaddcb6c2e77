// G1 multi-scalar multiplication (Pippenger bucket method) for BN254 on gfx950.
//
// Stands behind groth16/src/groth16.rs:88-96 (Groth16::prove -> bellman_ce::create_random_proof ->
// multiexp; the arithmetic itself is third-party).  Data layout = bellman's:
// bases n x 64 B affine (x, y), Fq in Montgomery form (R = 2^256), little-endian limbs; scalars
// n x 32 B canonical little-endian; result 64 B affine + infinity flag.
//
// Pipeline (all on the device, one stream):
//   1. digits: 16-bit windows (c = 16, 16 windows); histogram of (window, digit) keys with atomics
//   2. exclusive scan of the 2^20 counters -> bucket offsets
//   3. scatter point indices into bucket order (atomic cursors)
//   4. bucket accumulation: one lane per bucket, XYZZ += affine (8M + 2S per point), points
//      gathered by index (64 B each)
//   5. per-window reduction sum_k k*B_k: radix-16 hierarchy of (S, A) block summaries, 4 levels,
//      2^16 .. 2^4 lanes; the serial chain per lane is 47 point additions
//   6. Horner over the windows + conversion to affine (one lane)
// Field: 8 x 32-bit limbs, CIOS Montgomery multiplication on v_mad_u64_u32 (128 per product).
// Integer-ALU bound (about 10 Fq products per point and window); HBM traffic is 96 B per point.
#include "zk_internal.h"

namespace zk {

namespace bn254 {
constexpr int NL = 8;  // 32-bit limbs of Fq
// q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47 (alt_bn128 base field), R = 2^256
__host__ __device__ constexpr u32 FQ_Q(int i) {
    constexpr u32 q[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    return q[i];
}
__host__ __device__ constexpr u32 FQ_ONE(int i) {  // R mod q
    constexpr u32 r[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    return r[i];
}
constexpr u32 FQ_INV = 0xe4866389u;  // -q^-1 mod 2^32
__host__ __device__ constexpr u32 GEN_X(int i) { return FQ_ONE(i); }  // G = (1, 2)
__host__ __device__ constexpr u32 GEN_Y(int i) {  // 2R mod q
    constexpr u32 y[8] = {0x8b1e1b3au, 0xa6ba871bu, 0xeb8e167bu, 0x14f1d651u, 0xf0f28c58u, 0xccdd46deu, 0x340fbe5eu, 0x1c14ef83u};
    return y[i];
}
namespace {
#include "msm_impl.cuh"
}
}  // namespace bn254

namespace bls12_381 {
constexpr int NL = 12;
// q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab, R = 2^384
__host__ __device__ constexpr u32 FQ_Q(int i) {
    constexpr u32 q[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                           0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    return q[i];
}
__host__ __device__ constexpr u32 FQ_ONE(int i) {  // R mod q
    constexpr u32 r[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                           0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
    return r[i];
}
constexpr u32 FQ_INV = 0xfffcfffdu;  // -q^-1 mod 2^32
// the G1 generator of the BLS12-381 specification, Montgomery form
__host__ __device__ constexpr u32 GEN_X(int i) {
    constexpr u32 x[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u,
                           0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
    return x[i];
}
__host__ __device__ constexpr u32 GEN_Y(int i) {
    constexpr u32 y[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u,
                           0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};
    return y[i];
}
// 12-limb products stay out of line: inlined, one point addition is > 128 KB of code, beyond the reach of
// s_branch, and the relaxed long branches hipcc (ROCm 7.2) emits there hang the kernels
#undef FQ_MUL_ATTR
#define FQ_MUL_ATTR __noinline__
namespace {
#include "msm_impl.cuh"
}
}  // namespace bls12_381

void msm_g1_bn254_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st) {
    bn254::msm_g1_dev(d_bases, d_scalars, n, d_out, st);
}
void g1_bn254_mul_generator_dev(const u64* d_k, uint64_t n, void* d_bases, hipStream_t st) {
    bn254::g1_mul_generator_dev(d_k, n, d_bases, st);
}
void msm_g1_bls12_381_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st) {
    bls12_381::msm_g1_dev(d_bases, d_scalars, n, d_out, st);
}
void g1_bls12_381_mul_generator_dev(const u64* d_k, uint64_t n, void* d_bases, hipStream_t st) {
    bls12_381::g1_mul_generator_dev(d_k, n, d_bases, st);
}

}  // namespace zk
