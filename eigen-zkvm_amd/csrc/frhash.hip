// Scalar-field hashing for verificationHashType == "BN128" / "BLS12381" (the final STARK of every aggregation,
// test/stark_aggregation.sh:199-210) on gfx950; BN128 shown, the BLS12-381 files are textual twins:
//   Poseidon over the BN254 scalar field, t = 2..17      starky/src/poseidon_bn128_opt.rs:98-224
//   LinearHashBN128::hash_element_array                   starky/src/linearhash_bn128.rs:105-131
//   MerkleTreeBN128 (arity 16)                            starky/src/merklehash_bn128.rs:26-39, 196-239, 86-106
// A digest (ElementDigest<4, Fr>) holds the RAW limbs of an Fr, i.e. its Montgomery form a*2^256 mod r
// (digest.rs:45-53); that is the format of every node buffer here.
//
// Mapping: one lane = one permutation; the state (t <= 17 elements of 9 x 29-bit limbs) lives in the lane's
// private segment, the parameter tables (24 060 constants, converted once per device to the internal
// Montgomery form) in global memory behind wave-uniform addresses.  Integer-ALU bound: 5 457 Fr products per
// t = 17 permutation, 225 instructions each (fe29_impl.cuh).  Value bounds: round-boundary state < 2r; a
// column of the dense products sums 17 products (< 34r) and is brought back below 2r by one product with
// R' mod r; the sparse rounds' running columns are renormalised every 16 rounds (< 34r in between).
#include "zk_internal.h"
#include <cstdio>
#include <cstring>
#include <vector>

namespace zk {
namespace bn128fr {

constexpr int NL = 8;   // external Montgomery form: R = 2^256
constexpr int NR = 9;   // internal: R' = 2^261
constexpr u32 QINV29 = 0x0fffffffu;
#define ZK_FR_CONST(NAME, ...)                                                      \
    __host__ __device__ constexpr u32 NAME(int i) { constexpr u32 v[9] = {__VA_ARGS__}; return v[i]; }
// r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ZK_FR_CONST(Q29, 0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu)
ZK_FR_CONST(ONE29, 0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u)
ZK_FR_CONST(CIN29, 0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u)
ZK_FR_CONST(COUT29, 0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u)
ZK_FR_CONST(RRP29, 0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu, 0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au)  // R'^2 mod r
ZK_FR_CONST(Q2_29, 0x00000002u, 0x1e1f593fu, 0x1cb848a1u, 0x0fa121e6u, 0x0b0ba506u, 0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu)
ZK_FR_CONST(Q4_29, 0x00000004u, 0x1c3eb27eu, 0x19709143u, 0x1f4243cdu, 0x16174a0cu, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u)
ZK_FR_CONST(Q8_29, 0x00000008u, 0x187d64fcu, 0x12e12287u, 0x1e84879bu, 0x0c2e9419u, 0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u)
#undef ZK_FR_CONST
#define FH_NRP 56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68   // poseidon_bn128_opt.rs:62
#define FH_OUT_IDX 0                                                            // poseidon_bn128_opt.rs:80-83
#define FH_NAME "bn128"
#define FH_FN(name) bn128_##name
#include "frhash_impl.cuh"
#undef FH_NRP
#undef FH_OUT_IDX
#undef FH_NAME
#undef FH_FN
#undef FQ_MUL_ATTR
}  // namespace bn128fr

// ---- BLS12-381 scalar field: poseidon_bls12381_opt.rs (hash() returns state[1], :94-103), linearhash_bls12381.rs,
// ---- merklehash_bls12381.rs, transcript_bls12381.rs.  255-bit modulus in the same 9 x 29-bit limbs: q/R' < 2^-6,
// ---- so multiplicand bounds must satisfy A*B <= 68 (the permutation's largest is 34 * 1).
namespace bls12381fr {
constexpr int NL = 8;
constexpr int NR = 9;
constexpr u32 QINV29 = 0x1fffffffu;
#define ZK_FR_CONST(NAME, ...)                                                      \
    __host__ __device__ constexpr u32 NAME(int i) { constexpr u32 v[9] = {__VA_ARGS__}; return v[i]; }
// r = 52435875175126190479447740508185965837690552500527637822603658699938581184513
ZK_FR_CONST(Q29, 0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u)
ZK_FR_CONST(ONE29, 0x1fffffbau, 0x0000022fu, 0x1cb61180u, 0x0a4e5c00u, 0x0ee8b1a2u, 0x16e6aedfu, 0x1907f8bbu, 0x0853ddf7u, 0x004d043fu)
ZK_FR_CONST(CIN29, 0x1ffff72bu, 0x000046a7u, 0x1f5f3540u, 0x0ce3021cu, 0x118f3661u, 0x008176cbu, 0x054e487cu, 0x102e8190u, 0x001e092eu)
ZK_FR_CONST(COUT29, 0x1ffffffeu, 0x0000000fu, 0x00d20080u, 0x096ff400u, 0x04ff5588u, 0x07f7f65eu, 0x15be6631u, 0x0b3598a0u, 0x001824b1u)
ZK_FR_CONST(RRP29, 0x0a71b3c0u, 0x1d32207eu, 0x1663d999u, 0x1c5abc93u, 0x03b58c44u, 0x0be37438u, 0x0829f771u, 0x1660139eu, 0x0027fd91u)
ZK_FR_CONST(Q2_29, 0x00000002u, 0x1ffffff0u, 0x1f2dff7fu, 0x16900bffu, 0x1b00aa77u, 0x180809a1u, 0x0a4199ceu, 0x14ca675fu, 0x00e7db4eu)
ZK_FR_CONST(Q4_29, 0x00000004u, 0x1fffffe0u, 0x1e5bfeffu, 0x0d2017ffu, 0x160154efu, 0x10101343u, 0x1483339du, 0x0994cebeu, 0x01cfb69du)
ZK_FR_CONST(Q8_29, 0x00000008u, 0x1fffffc0u, 0x1cb7fdffu, 0x1a402fffu, 0x0c02a9deu, 0x00202687u, 0x0906673bu, 0x13299d7du, 0x039f6d3au)
#undef ZK_FR_CONST
#define FH_NRP 55, 55, 56, 56, 56, 56, 57, 57, 57, 57, 57, 57, 57, 57, 59, 59   // poseidon_bls12381_opt.rs:67
#define FH_OUT_IDX 1                                                            // poseidon_bls12381_opt.rs:94-103
#define FH_NAME "bls12381"
#define FH_FN(name) bls12381_##name
#include "frhash_impl.cuh"
}  // namespace bls12381fr

// the host entry points live in the field namespaces; zk_internal.h declares them in zk::
void bn128_load_constants(const char* path) { bn128fr::bn128_load_constants(path); }
void bn128_poseidon_dev(const u64* d_inp, uint64_t n, uint32_t n_in, const u64* d_init, uint32_t n_out, u64* d_out, hipStream_t st) { bn128fr::bn128_poseidon_dev(d_inp, n, n_in, d_init, n_out, d_out, st); }
uint64_t bn128_merkle_n_nodes(uint64_t h) { return bn128fr::bn128_merkle_n_nodes(h); }
void bn128_linearhash_rows_dev(const u64* r, uint32_t w, uint64_t h, u64* d, hipStream_t st) { bn128fr::bn128_linearhash_rows_dev(r, w, h, d, st); }
void bn128_merkelize_dev(const u64* r, uint32_t w, uint64_t h, u64* n, hipStream_t st) { bn128fr::bn128_merkelize_dev(r, w, h, n, st); }
void bls12381_load_constants(const char* path) { bls12381fr::bls12381_load_constants(path); }
void bls12381_poseidon_dev(const u64* d_inp, uint64_t n, uint32_t n_in, const u64* d_init, uint32_t n_out, u64* d_out, hipStream_t st) { bls12381fr::bls12381_poseidon_dev(d_inp, n, n_in, d_init, n_out, d_out, st); }
uint64_t bls12381_merkle_n_nodes(uint64_t h) { return bls12381fr::bls12381_merkle_n_nodes(h); }
void bls12381_linearhash_rows_dev(const u64* r, uint32_t w, uint64_t h, u64* d, hipStream_t st) { bls12381fr::bls12381_linearhash_rows_dev(r, w, h, d, st); }
void bls12381_merkelize_dev(const u64* r, uint32_t w, uint64_t h, u64* n, hipStream_t st) { bls12381fr::bls12381_merkelize_dev(r, w, h, n, st); }

}  // namespace zk
