/* ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the BN128-field hashing the reference uses when
 * verificationHashType == "BN128" (the final STARK of every aggregation, test/stark_aggregation.sh:199-210):
 *   Poseidon over the BN254 scalar field, t = 2..17      starky/src/poseidon_bn128_opt.rs:98-224
 *   LinearHashBN128 (3 Goldilocks words per Fr element)   starky/src/linearhash_bn128.rs:23-131
 *   MerkleTreeBN128 (arity 16)                            starky/src/merklehash_bn128.rs:26-39, 196-239, 86-106
 *   TranscriptBN128                                       starky/src/transcript_bn128.rs:22-132
 *   digest <-> scalar conventions                         starky/src/digest.rs:45-65, 162-175
 * Pinned by the reference's own known answers (tests/test_oracle_bn128.py): poseidon_bn128_opt.rs:233-300,
 * linearhash_bn128.rs:141-175, merklehash_bn128.rs:271-299.
 *
 * A digest (ElementDigest<4, Fr>) holds the RAW limbs of an Fr, i.e. its Montgomery form a*2^256 mod r
 * (digest.rs:45-53 from_scalar = into_raw_repr); this file speaks that format at its boundary ("raw").
 * Parameter tables: eigen-zkvm_amd/data/poseidon_bn128_constants.bin (a data table, one copy in the tree) (tools/gen_poseidon_bn128_constants.py). */
#define FH_X(name) orc_bn128_##name
#define FH_RMOD {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}
#define FH_R2 {1997599621687373223ULL, 6052339484930628067ULL, 10108755138030829701ULL, 150537098327114917ULL}   /* linearhash_bn128.rs:80-85 */
#define FH_RINV 0xc2e1f593efffffffULL
#define FH_NRP 56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68   /* poseidon_bn128_opt.rs:62 */
#define FH_OUT_IDX 0                                                            /* poseidon_bn128_opt.rs:80-83 */
#include "frhash_impl.h"
