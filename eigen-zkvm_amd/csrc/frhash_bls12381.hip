// The BLS12-381 scalar-field twin of frhash.hip (same body, frhash_impl.hip.h; a translation unit of its own so that the two compile side by side).
#include "zk_internal.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

namespace zk {
// ---- BLS12-381 scalar field: poseidon_bls12381_opt.rs (hash() returns state[1], :94-103), linearhash_bls12381.rs,
// ---- merklehash_bls12381.rs, transcript_bls12381.rs.  255-bit modulus in the same 9 x 29-bit limbs: q/R' < 2^-6,
// ---- so multiplicand bounds must satisfy A*B <= 68 (the permutation's largest is 34 * 1).
namespace bls12381fr {
#define ZK_FR29_FIELD 381
#include "fr29_consts.hip.h"
#define FH_NRP 55, 55, 56, 56, 56, 56, 57, 57, 57, 57, 57, 57, 57, 57, 59, 59   // poseidon_bls12381_opt.rs:67
#define FH_OUT_IDX 1                                                            // poseidon_bls12381_opt.rs:94-103
#undef FH_AB_LIMIT
#define FH_AB_LIMIT 68u
#define FH_NAME "bls12381"
#define FH_FN(name) bls12381_##name
#include "frhash_impl.hip.h"
}  // namespace bls12381fr

void bls12381_load_constants(const char* path) { bls12381fr::bls12381_load_constants(path); }
std::string bls12381_tables_selfcheck(const char* path) { return bls12381fr::bls12381_tables_selfcheck(path); }
void bls12381_poseidon_dev(const u64* d_inp, uint64_t n, uint32_t n_in, const u64* d_init, uint32_t n_out, u64* d_out, hipStream_t st) { bls12381fr::bls12381_poseidon_dev(d_inp, n, n_in, d_init, n_out, d_out, st); }
uint64_t bls12381_merkle_n_nodes(uint64_t h) { return bls12381fr::bls12381_merkle_n_nodes(h); }
void bls12381_linearhash_rows_dev(const u64* r, uint32_t w, uint64_t h, u64* d, hipStream_t st) { bls12381fr::bls12381_linearhash_rows_dev(r, w, h, d, st); }
void bls12381_merkelize_dev(const u64* r, uint32_t w, uint64_t h, u64* n, hipStream_t st) { bls12381fr::bls12381_merkelize_dev(r, w, h, n, st); }

}  // namespace zk
