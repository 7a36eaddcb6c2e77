// Multi-scalar multiplication (Pippenger bucket method) on G1 and G2 of BN254 and BLS12-381 for gfx950.
//
// Stands behind groth16/src/groth16.rs:88-96 (Groth16::prove -> bellman_ce::create_random_proof ->
// multiexp; the arithmetic itself is third-party).  Data layout = bellman's / pairing_ce's:
// bases n x 64 B (BN254 G1) affine (x, y), Fq in Montgomery form (R = 2^256), little-endian limbs; scalars
// n x 32 B canonical little-endian; result affine + infinity flag.  96 B points for BLS12-381, twice that for G2.
//
// Pipeline (msm_impl.hip.h; all on the device, one stream):
//   0. bases: external 32-bit-limb Montgomery form -> internal 29-bit limbs (one product per coordinate)
//   1. bucket sort of the 16 n (point, window) pairs by key = window * 2^16 + digit (c = 16): two LDS-histogram
//      partition passes, no device-scope atomics
//   2. bucket ids ordered by decreasing size (counting sort), so that the lanes of a wave finish together
//   3. bucket accumulation: one lane per bucket, XYZZ += affine (8M + 2S per point), points gathered by index
//   4. per-window reduction sum_k k*B_k: radix-16 hierarchy of (S, A) block summaries, 4 levels,
//      2^16 .. 2^4 lanes; the serial chain per lane is 47 point additions
//   5. Horner over the windows + conversion to affine (one lane)
// Field: 29-bit limbs with 64-bit column accumulators, lazily reduced (fe29_impl.hip.h); Fq2 on top of it for G2.
// Integer-ALU bound (about 10 Fq products per point and window); HBM traffic is 96 B per point.
#include "zk_internal.h"

// fe29_impl.hip.h: the sums keep squaring as fe_mul(a, a).  The dedicated squaring of the scalar-field hashes was measured here
// (tools/gpu_sqr_ab.sh, profiles/r05/sqr_ab_raw.txt): G1 +-1 %, BN254 G2 3 % slower -- the accumulation kernels sit at a register edge.
#define ZK_FE_SQR_PLAIN 1

namespace zk {

namespace bn254 {
constexpr int NL = 8;  // 32-bit limbs of Fq
// q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47 (alt_bn128 base field); external R = 2^256
constexpr int NR = 9;  // 29-bit limbs of the internal representation, R' = 2^261
constexpr u32 QINV29 = 0x04866389u;  // -q^-1 mod 2^29
__host__ __device__ constexpr u32 Q29(int i) {
    constexpr u32 v[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    return v[i];
}
__host__ __device__ constexpr u32 ONE29(int i) {
    constexpr u32 v[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    return v[i];
}
__host__ __device__ constexpr u32 CIN29(int i) {
    constexpr u32 v[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
    return v[i];
}
__host__ __device__ constexpr u32 COUT29(int i) {
    constexpr u32 v[9] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
    return v[i];
}
__host__ __device__ constexpr u32 Q2_29(int i) {
    constexpr u32 v[9] = {0x10f9fa8eu, 0x0208c16du, 0x18e5469eu, 0x05aa45a1u, 0x0b0bb2f0u, 0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu};
    return v[i];
}
__host__ __device__ constexpr u32 Q4_29(int i) {
    constexpr u32 v[9] = {0x01f3f51cu, 0x041182dbu, 0x11ca8d3cu, 0x0b548b43u, 0x161765e0u, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u};
    return v[i];
}
__host__ __device__ constexpr u32 Q8_29(int i) {
    constexpr u32 v[9] = {0x03e7ea38u, 0x082305b6u, 0x03951a78u, 0x16a91687u, 0x0c2ecbc0u, 0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u};
    return v[i];
}
__host__ __device__ constexpr u32 RRP29(int i) {  // R'^2 mod q: canonical integers -> internal form (key files store canonical coordinates)
    constexpr u32 v[9] = {0x059bac10u, 0x0d1503a3u, 0x018016b8u, 0x10ab0ca8u, 0x02632639u, 0x02c0169fu, 0x169bfd53u, 0x11869d4cu, 0x002a11a6u};
    return v[i];
}
namespace g1 {
__host__ __device__ constexpr u32 GEN_X(int i) {  // G = (1, 2), external Montgomery form: R mod q
    constexpr u32 r[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    return r[i];
}
__host__ __device__ constexpr u32 GEN_Y(int i) {  // 2R mod q
    constexpr u32 y[8] = {0x8b1e1b3au, 0xa6ba871bu, 0xeb8e167bu, 0x14f1d651u, 0xf0f28c58u, 0xccdd46deu, 0x340fbe5eu, 0x1c14ef83u};
    return y[i];
}
}  // namespace g1
namespace g1half {   // the same group with 128-bit scalars (8 windows, 4 words apart): the sum behind the endomorphism split of g1
using g1::GEN_X; using g1::GEN_Y;
#define MSM_N_WIN 8
#define MSM_SC_WORDS 4
namespace {
#include "msm_impl.hip.h"
}
}  // namespace g1half
namespace g1 {
// phi(x, y) = (beta x, y) = [lambda](x, y) on y^2 = x^3 + 3: beta = 2203960485148121921418603742825762020974279258880205651966 (beta^3 = 1 in
// Fq), lambda = 4407920970296243842393367215006156084916469457145843978461; lattice basis a1 = b2 = 9931322734385697763,
// -b1 = 147946756881789319000765030803803410728, a2 = 147946756881789319010696353538189108491 (tools/glv_constants.py derives and checks them)
#define GLV_BETA_STD 0xd782e155u, 0x71930c11u, 0xffbe3323u, 0xa6bb947cu, 0xd4741444u, 0xaa303344u, 0x26594943u, 0x2c3b3f0du
#define GLV_G1 0xc7e0b3d7u, 0xd91d232eu, 0x00000002u
#define GLV_G2 0x391eb18du, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x00000002u
#define GLV_A1 0x94d213e3u, 0x89d32568u
#define GLV_A2 0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u
#define GLV_NB1 0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u
#define GLV_B2 0x94d213e3u, 0x89d32568u
#define MSM_GLV g1half
namespace {
#include "msm_impl.hip.h"
}
#undef MSM_GLV
#undef GLV_BETA_STD
#undef GLV_G1
#undef GLV_G2
#undef GLV_A1
#undef GLV_A2
#undef GLV_NB1
#undef GLV_B2
}  // namespace g1
namespace g2 {   // the twist y^2 = x^3 + 3/(9 + u) over Fq2 = Fq[u]/(u^2 + 1); generator of EIP-197, x = c0 + c1 u
__host__ __device__ constexpr u32 GEN_X(int i) {
    constexpr u32 v[16] = {0x02bc2026u, 0x8e83b5d1u, 0x497b0172u, 0xdceb1935u, 0x97811adfu, 0xfbb82647u, 0xaf96503bu, 0x19573841u,
                           0xa84c6140u, 0xafb4737du, 0x5802d8c4u, 0x6043dd5au, 0x52a02f86u, 0x09e950fcu, 0x3aea7b6bu, 0x14fef083u};
    return v[i];
}
__host__ __device__ constexpr u32 GEN_Y(int i) {
    constexpr u32 v[16] = {0x886be9f6u, 0x619dfa9du, 0xf59e9b78u, 0xfe7fd297u, 0x231b7dfeu, 0xff9e1a62u, 0xae9e4206u, 0x28fd7eebu,
                           0xc71856eeu, 0x64095b56u, 0x327d3cbbu, 0xdc57f922u, 0x33351076u, 0x55f935beu, 0x93fd6482u, 0x0da4a0e6u};
    return v[i];
}
// Code-size hazard: with everything inlined, as for G1, a G2 point addition built from six-product Fq2 multiplications
// was > 128 KB of code, beyond the reach of s_branch, and kernels of that size built by hipcc (ROCm 7.2) did not
// terminate on the device (seen twice: 12 x 32-bit limbs inlined for BLS12-381 G1, and this).  With the two-reduction
// Fq2 product (fe_mul2) a BN254 addition is ~70 KB: the Fq2 products are inlined into the point formulas, the hot
// mixed addition into the accumulate kernel, and the cold formulas (pt_add, pt_dbl, pt_dbl_aff) stay real functions.
// BLS12-381 (14 limbs, 2.4 x the code) keeps cf_mul / cf_sqr as functions.
#define MSM_G2_INLINE_CF   // 9 x 29-bit limbs: the Fq2 products are small enough to inline; the rare point formulas stay out of line
#define MSM_G2
namespace {
#include "msm_impl.hip.h"
}
#undef MSM_G2_INLINE_CF
#undef CF_MUL_ATTR
#undef PT_COLD_ATTR
#undef MSM_G2
#undef FQ_MUL_ATTR
}  // namespace g2
}  // namespace bn254

namespace bls12_381 {
constexpr int NL = 12;
// q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab; external R = 2^384
constexpr int NR = 14;  // 29-bit limbs of the internal representation, R' = 2^406
constexpr u32 QINV29 = 0x1ffcfffdu;  // -q^-1 mod 2^29
__host__ __device__ constexpr u32 Q29(int i) {
    constexpr u32 v[14] = {0x1fffaaabu, 0x0ff7ffffu, 0x14ffffeeu, 0x17fffd62u, 0x0f6241eau, 0x09507b58u, 0x0afd9cc3u, 0x109e70a2u, 0x1764774bu, 0x121a5d66u, 0x12c6e9edu, 0x12ffcd34u, 0x00111ea3u, 0x0000000du};
    return v[i];
}
__host__ __device__ constexpr u32 ONE29(int i) {
    constexpr u32 v[14] = {0x03a9fb84u, 0x0ba00690u, 0x071288f1u, 0x0f59bcc5u, 0x126cb614u, 0x0585bf36u, 0x1b85ac3du, 0x1cf856fau, 0x1891ecbdu, 0x1a7eec05u, 0x155a88f0u, 0x0741ac6du, 0x1317c30fu, 0x00000009u};
    return v[i];
}
__host__ __device__ constexpr u32 CIN29(int i) {
    constexpr u32 v[14] = {0x1fddebbdu, 0x1a4f5474u, 0x0291f399u, 0x14d03b3cu, 0x0f6cad2cu, 0x1b4cabcau, 0x1592827cu, 0x021c6ac7u, 0x1ec52a84u, 0x16fd5ec4u, 0x0c960da6u, 0x0fd2af6bu, 0x13263591u, 0x0000000bu};
    return v[i];
}
__host__ __device__ constexpr u32 COUT29(int i) {
    constexpr u32 v[14] = {0x0002fffdu, 0x10480000u, 0x0300009du, 0x08001788u, 0x158baebfu, 0x0c2ba9e3u, 0x1d157d22u, 0x0a6e0a4au, 0x0d77ce58u, 0x1d12b763u, 0x1701c6a5u, 0x1501c926u, 0x1f65ec3fu, 0x0000000au};
    return v[i];
}
__host__ __device__ constexpr u32 Q2_29(int i) {
    constexpr u32 v[14] = {0x1fff5556u, 0x1fefffffu, 0x09ffffdcu, 0x0ffffac5u, 0x1ec483d5u, 0x12a0f6b0u, 0x15fb3986u, 0x013ce144u, 0x0ec8ee97u, 0x0434bacdu, 0x058dd3dbu, 0x05ff9a69u, 0x00223d47u, 0x0000001au};
    return v[i];
}
__host__ __device__ constexpr u32 Q4_29(int i) {
    constexpr u32 v[14] = {0x1ffeaaacu, 0x1fdfffffu, 0x13ffffb9u, 0x1ffff58au, 0x1d8907aau, 0x0541ed61u, 0x0bf6730du, 0x0279c289u, 0x1d91dd2eu, 0x0869759au, 0x0b1ba7b6u, 0x0bff34d2u, 0x00447a8eu, 0x00000034u};
    return v[i];
}
__host__ __device__ constexpr u32 Q8_29(int i) {
    constexpr u32 v[14] = {0x1ffd5558u, 0x1fbfffffu, 0x07ffff73u, 0x1fffeb15u, 0x1b120f55u, 0x0a83dac3u, 0x17ece61au, 0x04f38512u, 0x1b23ba5cu, 0x10d2eb35u, 0x16374f6cu, 0x17fe69a4u, 0x0088f51cu, 0x00000068u};
    return v[i];
}
__host__ __device__ constexpr u32 RRP29(int i) {  // R'^2 mod q
    constexpr u32 v[14] = {0x15bef7aeu, 0x1031cd0eu, 0x02dd93e8u, 0x09226323u, 0x0e6e2cd2u, 0x11684daau, 0x1170e5dbu, 0x088e25b1u, 0x1b366399u, 0x1c536f47u, 0x0d1f9cbcu, 0x0278b67fu, 0x1ea66a2bu, 0x0000000cu};
    return v[i];
}
namespace g1 {
// the G1 generator of the BLS12-381 specification, external Montgomery form (R = 2^384)
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
}  // namespace g1
namespace g1half {   // 128-bit scalars, 8 windows: the sum behind the endomorphism split of g1
using g1::GEN_X; using g1::GEN_Y;
#define MSM_N_WIN 8
#define MSM_SC_WORDS 4
namespace {
#include "msm_impl.hip.h"
}
}  // namespace g1half
namespace g1 {
// phi(x, y) = (beta x, y) = [lambda](x, y) on y^2 = x^3 + 4 with lambda = z^2 - 1 = 0xac45a4010001a40200000000ffffffff (z the curve parameter):
// lambda^2 + lambda + 1 = 0 mod r, so k = k1 + k2 lambda by division; beta =
// 4002409555221667392624310435006688643935503118305586438271171395842971157480381377015405980053539358417135540939436
#define GLV_BETA_STD 0x8671f071u, 0xcd03c9e4u, 0x1fcda5d2u, 0x5dab2246u, 0xd3851b95u, 0x587042afu, 0x01bacb9eu, 0x8eb60ebeu, 0x83d050d2u, 0x03f97d6eu, 0x54638741u, 0x18f02065u
#define GLV_LAMBDA 0xffffffffu, 0x00000000u, 0x0001a402u, 0xac45a401u
#define GLV_G 0xf6cfee30u, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u, 0x00000001u
#define GLV_R 0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u
#define MSM_GLV g1half
namespace {
#include "msm_impl.hip.h"
}
#undef MSM_GLV
#undef GLV_BETA_STD
#undef GLV_LAMBDA
#undef GLV_G
#undef GLV_R
}  // namespace g1
namespace g2 {   // the twist y^2 = x^3 + 4(1 + u) over Fq2 = Fq[u]/(u^2 + 1); G2 generator of the BLS12-381 specification
__host__ __device__ constexpr u32 GEN_X(int i) {
    constexpr u32 v[24] = {0x02940a10u, 0xf5f28fa2u, 0x87b4961au, 0xb3f5fb26u, 0x3e2ae580u, 0xa1a893b5u, 0x1a3caee9u, 0x9894999du, 0x1863366bu, 0x6f67b763u, 0x4350bcd7u, 0x05819192u,
                           0x9e23f606u, 0xa5a9c075u, 0xbccd60c3u, 0xaaa0c59du, 0xe2867806u, 0x3bb17e18u, 0x8541b367u, 0x1b1ab6ccu, 0xf2158547u, 0xc2b6ed0eu, 0x7360edf3u, 0x11922a09u};
    return v[i];
}
__host__ __device__ constexpr u32 GEN_Y(int i) {
    constexpr u32 v[24] = {0x60494c4au, 0x4c730af8u, 0x5e369c5au, 0x597cfa1fu, 0xaa0a635au, 0xe7e6856cu, 0x6e0d495fu, 0xbbefb5e9u, 0xf0ef25a2u, 0x07d3a975u, 0x7e80dae5u, 0x0083fd8eu,
                           0xdf64b05du, 0xadc0fc92u, 0x2b1461dcu, 0x18aa270au, 0x3be4eba0u, 0x86adac6au, 0xc93da33au, 0x79495c4eu, 0xa43ccaedu, 0xe7175850u, 0x63de1bf2u, 0x0b2bc2a1u};
    return v[i];
}
// Code-size hazard: with everything inlined, as for G1, a G2 point addition built from six-product Fq2 multiplications
// was > 128 KB of code, beyond the reach of s_branch, and kernels of that size built by hipcc (ROCm 7.2) did not
// terminate on the device (seen twice: 12 x 32-bit limbs inlined for BLS12-381 G1, and this).  With the two-reduction
// Fq2 product (fe_mul2) a BN254 addition is ~70 KB: the Fq2 products are inlined into the point formulas, the hot
// mixed addition into the accumulate kernel, and the cold formulas (pt_add, pt_dbl, pt_dbl_aff) stay real functions.
// BLS12-381 (14 limbs, 2.4 x the code) keeps cf_mul / cf_sqr as functions.
#define MSM_G2
namespace {
#include "msm_impl.hip.h"
}
#undef CF_MUL_ATTR
#undef PT_COLD_ATTR
#undef MSM_G2
#undef FQ_MUL_ATTR
}  // namespace g2
}  // namespace bls12_381

void msm_g1_bn254_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st) {
    bn254::g1::msm_g1_dev(d_bases, d_scalars, n, d_out, st);
}
void g1_bn254_mul_generator_dev(const u64* d_k, uint64_t n, void* d_bases, hipStream_t st) {
    bn254::g1::g1_mul_generator_dev(d_k, n, d_bases, st);
}
void msm_g1_bls12_381_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st) {
    bls12_381::g1::msm_g1_dev(d_bases, d_scalars, n, d_out, st);
}
void g1_bls12_381_mul_generator_dev(const u64* d_k, uint64_t n, void* d_bases, hipStream_t st) {
    bls12_381::g1::g1_mul_generator_dev(d_k, n, d_bases, st);
}

// G2: points n x 4 coordinates-words (x.c0, x.c1, y.c0, y.c1), i.e. 128 B (BN254) / 192 B (BLS12-381) each
void fq_bn254_canon_to_mont_dev(void* d, uint64_t n, hipStream_t st) { bn254::g1::fq_canon_to_mont_dev(d, n, st); }
void fq_bn254_mont_to_canon_dev(void* d, uint64_t n, hipStream_t st) { bn254::g1::fq_mont_to_canon_dev(d, n, st); }
void fq_bls12_381_canon_to_mont_dev(void* d, uint64_t n, hipStream_t st) { bls12_381::g1::fq_canon_to_mont_dev(d, n, st); }
void fq_bls12_381_mont_to_canon_dev(void* d, uint64_t n, hipStream_t st) { bls12_381::g1::fq_mont_to_canon_dev(d, n, st); }
void msm_g2_bn254_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st) { bn254::g2::msm_g1_dev(d_bases, d_scalars, n, d_out, st); }
void g2_bn254_mul_generator_dev(const u64* d_k, uint64_t n, void* d_bases, hipStream_t st) { bn254::g2::g1_mul_generator_dev(d_k, n, d_bases, st); }
void msm_g2_bls12_381_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st) { bls12_381::g2::msm_g1_dev(d_bases, d_scalars, n, d_out, st); }
void g2_bls12_381_mul_generator_dev(const u64* d_k, uint64_t n, void* d_bases, hipStream_t st) { bls12_381::g2::g1_mul_generator_dev(d_k, n, d_bases, st); }

// window tables (fixed bases)
#define ZK_MSM_FIXED(NAME, NS)                                                                                              \
    size_t msm_##NAME##_fixed_table_bytes(uint64_t n) { return NS::msm_fixed_table_bytes(n); }                              \
    void msm_##NAME##_fixed_prepare_dev(const void* b, uint64_t n, void* t, hipStream_t st) { NS::msm_fixed_prepare_dev(b, n, t, st); } \
    void msm_##NAME##_fixed_dev(const void* t, uint64_t tn, uint64_t off, const void* s, uint64_t n, void* o, hipStream_t st) { NS::msm_fixed_dev(t, tn, off, s, n, o, st); }
ZK_MSM_FIXED(g1_bn254, bn254::g1)
ZK_MSM_FIXED(g2_bn254, bn254::g2)
ZK_MSM_FIXED(g1_bls12_381, bls12_381::g1)
ZK_MSM_FIXED(g2_bls12_381, bls12_381::g2)
#undef ZK_MSM_FIXED

}  // namespace zk
