/* ORACLE -- TEST INFRASTRUCTURE ONLY.  BLS12-381 G1 instance of the MSM restatement (ec_impl.h).
 *
 * The reference selects this curve with `--curve BLS12381` (zkit/src/main.rs, groth16/src/api.rs) and
 * reaches its arithmetic through the same out-of-tree crates as BN254 (pairing_ce 0.24.2 `bls12_381`,
 * bellman_ce multiexp; see ec.c).  No known-answer vector in the reference: pinned by algebra and by
 * the standard constants -- q (381 bits), y^2 = x^3 + 4, the G1 generator of the BLS12-381
 * specification (on-curve and [r]G = infinity are checked in tests/test_oracle_msm.py).  Layout as
 * pairing_ce keeps it: Fq = 6 little-endian u64 limbs in Montgomery form (R = 2^384), points 96 B
 * affine x || y, scalars 4 x u64 canonical. */
#define EC_NL 6
#define EC_Q {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL}
#define EC_R1 {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL} /* 2^384 mod q */
#define EC_R2 {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL, 0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL} /* 2^768 mod q */
#define EC_QINV 0x89f3fffcfffcfffdULL /* -q^-1 mod 2^64 */
#define EC_B 4
#define EC_GX {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL, 0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL}
#define EC_GY {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL, 0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL}
#define EC_B2C0 {4, 0, 0, 0, 0, 0}   /* b' = 4(1 + u) */
#define EC_B2C1 {4, 0, 0, 0, 0, 0}
#define EC_G2X0 {0xd48056c8c121bdb8ULL, 0x0bac0326a805bbefULL, 0xb4510b647ae3d177ULL, 0xc6e47ad4fa403b02ULL, 0x260805272dc51051ULL, 0x024aa2b2f08f0a91ULL}   /* G2 generator of the BLS12-381 specification */
#define EC_G2X1 {0xe5ac7d055d042b7eULL, 0x334cf11213945d57ULL, 0xb5da61bbdc7f5049ULL, 0x596bd0d09920b61aULL, 0x7dacd3a088274f65ULL, 0x13e02b6052719f60ULL}
#define EC_G2Y0 {0xe193548608b82801ULL, 0x923ac9cc3baca289ULL, 0x6d429a695160d12cULL, 0xadfd9baa8cbdd3a7ULL, 0x8cc9cdc6da2e351aULL, 0x0ce5d527727d6e11ULL}
#define EC_G2Y1 {0xaaa9075ff05f79beULL, 0x3f370d275cec1da1ULL, 0x267492ab572e99abULL, 0xcb3e287e85a763afULL, 0x32acd2b02bc28b99ULL, 0x0606c4a02ea734ccULL}
#define EC_X(name) orc_bls12_381_##name
#include "ec_impl.h"
