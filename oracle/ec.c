/* ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of G1 multi-scalar multiplication.
 *
 * The reference reaches MSM only through a third-party dependency that is NOT under
 * /root/reference: bellman_ce 0.3.2 (git matter-labs/bellman@416f79d3, via franklin-crypto
 * 0.0.5@9e3c2a12; Cargo.lock:668-670, 2259-2261), curves from pairing_ce 0.24.2
 * (Cargo.lock:3387-3389); call site groth16/src/groth16.rs:93 (create_random_proof).  The reference
 * holds no MSM known-answer vector and its proofs are randomised, so MSM PARITY IS UNPINNED by the
 * reference.  This file restates the published algorithm (Pippenger bucket method over the
 * short-Weierstrass curve y^2 = x^3 + 3, alt_bn128 / BN254, generator (1, 2)) with the data layout
 * bellman uses: bases = affine (x, y), Fq in Montgomery form (R = 2^256), 4 little-endian u64 limbs
 * each; scalars = canonical 256-bit little-endian FrRepr.  It is pinned by algebra instead:
 * sum s_i [k_i]G == [sum s_i k_i mod r]G (tests/test_oracle_msm.py), on-curve checks and the
 * standard alt_bn128 constants.  Jacobian coordinates here, XYZZ on the GPU: different formulas,
 * same unique affine result.
 */
#define EC_NL 4
#define EC_Q {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}
#define EC_R1 {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL} /* 2^256 mod q */
#define EC_R2 {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL} /* 2^512 mod q */
#define EC_QINV 0x87d20782e4866389ULL /* -q^-1 mod 2^64 */
#define EC_B 3
#define EC_GX {1, 0, 0, 0}
#define EC_GY {2, 0, 0, 0}
#define EC_B2C0 {0x3267e6dc24a138e5ULL, 0xb5b4c5e559dbefa3ULL, 0x81be18991be06ac3ULL, 0x2b149d40ceb8aaaeULL}   /* b' = 3/(9 + u) */
#define EC_B2C1 {0xe4a2bd0685c315d2ULL, 0xa74fa084e52d1852ULL, 0xcd2cafadeed8fdf4ULL, 0x009713b03af0fed4ULL}
#define EC_G2X0 {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL}   /* G2 generator of EIP-197 */
#define EC_G2X1 {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL}
#define EC_G2Y0 {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL}
#define EC_G2Y1 {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL}
#define EC_X(name) orc_bn254_##name
#include "ec_impl.h"

/* historical names used by the tests */
void orc_fq_mul(const uint64_t a[4], const uint64_t b[4], uint64_t r[4]) { orc_bn254_fq_mul(a, b, r); }
void orc_fq_from_mont(const uint64_t a[4], uint64_t r[4]) { orc_bn254_fq_from_mont(a, r); }
void orc_fq_to_mont(const uint64_t a[4], uint64_t r[4]) { orc_bn254_fq_to_mont(a, r); }
