/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Groth16 scalar-field work over the BLS12-381 scalar field (see groth16_impl.h). */
#define G16_X(name) orc_g16_bls12_381_##name
#define G16_RMOD 0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL
#define G16_R2 0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL
#define G16_RINV 0xfffffffeffffffffULL
#define G16_S 32
#define G16_ROOT 0xb9b58d8c5f0e466aULL, 0x5b1b4c801819d7ecULL, 0x0af53ae352a31e64ULL, 0x5bf3adda19e9b27bULL   /* 7^((r-1)/2^32) * 2^256 */
#include "groth16_impl.h"
