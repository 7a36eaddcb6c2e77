/* ORACLE -- TEST INFRASTRUCTURE ONLY.  BLS12-381 scalar-field instance of the hashing restatement (frhash_impl.h):
 * verificationHashType == "BLS12381" (`--curve BLS12381`, test/stark_aggregation.sh).  Follows
 *   starky/src/poseidon_bls12381_opt.rs:104-230 (same schedule as BN128; n_rounds_p :67; hash() returns state[1], :94-103)
 *   starky/src/linearhash_bls12381.rs, merklehash_bls12381.rs, transcript_bls12381.rs (textual twins of the BN128 files)
 * Pinned by the reference's known answers in tests/test_oracle_bn128.py: poseidon_bls12381_opt.rs:236-311,
 * linearhash_bls12381.rs:141-193, merklehash_bls12381.rs:274-300. */
#define FH_X(name) orc_bls12381_##name
#define FH_RMOD {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}
#define FH_R2 {14526898881837571181ULL, 3129137299524312099ULL, 419701826671360399ULL, 524908885293268753ULL}   /* linearhash_bls12381.rs:79-84 */
#define FH_RINV 0xfffffffeffffffffULL
#define FH_NRP 55, 55, 56, 56, 56, 56, 57, 57, 57, 57, 57, 57, 57, 57, 59, 59   /* poseidon_bls12381_opt.rs:67 */
#define FH_OUT_IDX 1                                                            /* poseidon_bls12381_opt.rs:94-103 */
#include "frhash_impl.h"
