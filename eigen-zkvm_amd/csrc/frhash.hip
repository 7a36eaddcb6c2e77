// Scalar-field hashing for verificationHashType == "BN128" / "BLS12381" (the final STARK of every aggregation,
// test/stark_aggregation.sh:199-210) on gfx950; BN128 shown, the BLS12-381 files are textual twins:
//   Poseidon over the BN254 scalar field, t = 2..17      starky/src/poseidon_bn128_opt.rs:98-224
//   LinearHashBN128::hash_element_array                   starky/src/linearhash_bn128.rs:105-131
//   MerkleTreeBN128 (arity 16)                            starky/src/merklehash_bn128.rs:26-39, 196-239, 86-106
// A digest (ElementDigest<4, Fr>) holds the RAW limbs of an Fr, i.e. its Montgomery form a*2^256 mod r
// (digest.rs:45-53); that is the format of every node buffer here.
//
// Mapping, by how many permutations a launch holds (frhash_impl.hip.h): one lane per permutation with the state
// in registers where throughput binds (leaves on more than 4096 rows -- one sponge step, or the full steps of a wide row and then its last --,
// levels above 16 384 parents): since round 6 every linear step of these kernels (the dense layers M, P and the row / column products of the
// sparse rounds) runs on the matrix pipe from digit tables built on the host (fr_mfma.hip.h), the vector pipe keeps the S-boxes;
// eight lanes per permutation for levels of 2 561..16 384 parents, one wave per permutation (sparse rounds as a linear recurrence over
// the lanes) where a chain's latency binds (small levels, few wide rows, the transcript).  The parameter tables (24 060 constants,
// converted once per device to the internal Montgomery form, the cooperative form's coefficient tables built from them on the
// device, the matrix-pipe fragments of t = 3..17: ~20 MB per field) live in global memory.  Integer-ALU bound: 325 k vector instructions per
// one-lane t = 17 permutation (1 025 k before round 6), 126 k of them the 204 S-boxes (fe29_impl.hip.h).  Value bounds are stated where the sums are formed.
#include "zk_internal.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

namespace zk {
namespace bn128fr {

#define ZK_FR29_FIELD 254
#include "fr29_consts.hip.h"
#define FH_NRP 56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68   // poseidon_bn128_opt.rs:62
#define FH_OUT_IDX 0                                                            // poseidon_bn128_opt.rs:80-83
#define FH_AB_LIMIT 168u
#define FH_NAME "bn128"
#define FH_FN(name) bn128_##name
#include "frhash_impl.hip.h"
#undef FH_NRP
#undef FH_OUT_IDX
#undef FH_NAME
#undef FH_FN
#undef FQ_MUL_ATTR
}  // namespace bn128fr

// the host entry points live in the field namespace; zk_internal.h declares them in zk::.  (The BLS12-381 twin is a translation unit
// of its own, frhash_bls12381.hip: the two take over a minute each to compile.)
void bn128_load_constants(const char* path) { bn128fr::bn128_load_constants(path); }
std::string bn128_tables_selfcheck(const char* path) { return bn128fr::bn128_tables_selfcheck(path); }
void bn128_poseidon_dev(const u64* d_inp, uint64_t n, uint32_t n_in, const u64* d_init, uint32_t n_out, u64* d_out, hipStream_t st) { bn128fr::bn128_poseidon_dev(d_inp, n, n_in, d_init, n_out, d_out, st); }
uint64_t bn128_merkle_n_nodes(uint64_t h) { return bn128fr::bn128_merkle_n_nodes(h); }
void bn128_linearhash_rows_dev(const u64* r, uint32_t w, uint64_t h, u64* d, hipStream_t st) { bn128fr::bn128_linearhash_rows_dev(r, w, h, d, st); }
void bn128_merkelize_dev(const u64* r, uint32_t w, uint64_t h, u64* n, hipStream_t st) { bn128fr::bn128_merkelize_dev(r, w, h, n, st); }

}  // namespace zk
