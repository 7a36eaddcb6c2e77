// Internal plumbing shared by the HIP translation units of libzkgpu (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <stdexcept>
#include <vector>
#include "gl.hip.h"

struct zk_merkle;
struct zk_transcript;
namespace zk {

// ---- error handling: C ABI returns int status, message kept per thread (include/zkgpu.h) ----
void set_error(const std::string& msg);
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

#define ZK_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            throw zk::Error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" +     \
                            __FILE__ + ":" + std::to_string(__LINE__) + ")");                     \
    } while (0)

#define ZK_REQUIRE(cond, msg)                                                                     \
    do { if (!(cond)) throw zk::Error(std::string(msg)); } while (0)

// ---- device buffers ----
// Size-keyed free list in front of hipMalloc/hipFree (capi.hip): a prover allocates the same
// multi-GB sections for every proof, and hipFree/hipMalloc of such buffers costs hundreds of ms and
// synchronises the device.  Reuse is ordered across streams: every host thread keeps the set of streams it has
// issued on (`on_stream`); pool_free records an event on the null stream and on each stream of the freeing thread,
// and pool_alloc makes the calling thread's current stream wait for the events of the block it hands out, so a
// block freed with kernels still in flight on stream A is never written early by stream B -- while provers running
// on different threads and streams never wait for each other.  With a single stream in use (the common case) no
// event is needed: reuse is stream ordered.
void* pool_alloc(size_t bytes, bool host_wait = false);   // host_wait: drain the previous users on the host (the block leaves the library)
void h2d_sync(void* d, const void* h, size_t n);            // copies of pooled buffers, ordered on the current stream and waited for
void d2h_sync(void* h, const void* d, size_t n);
void pool_free(void* p);
void pool_defer_begin();   // frees of this thread are collected until pool_defer_flush() stamps them with one set of events
void pool_defer_flush();
void pool_trim();  // hipFree everything cached
// registers `st` as a stream the library works on and makes it the calling thread's current stream (the one
// pool_alloc orders reuse against); returns st.  Every entry point that takes a stream goes through it.
hipStream_t on_stream(hipStream_t st);
hipStream_t cur_stream();
// Scope of one C-ABI call: a call from outside the library starts on the null stream, a call the prover makes on its own
// entry points inherits the prover's current stream; either way the caller's current stream is back when the call returns.
void bind_device() noexcept;           // the calling thread onto the GPU zk_init selected (HIP's current device is per thread)
struct CallScope { hipStream_t saved; CallScope(); ~CallScope(); CallScope(const CallScope&) = delete; CallScope& operator=(const CallScope&) = delete; };
void forget_stream(hipStream_t st);  // call before destroying a registered stream, or when a side stream's work has been waited for
void on_side_stream(hipStream_t ss); // a helper stream inside one call (see capi.hip); pairs with forget_stream

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) pool_free(p); }
    void reserve(size_t n) {  // grow-only workspace
        if (n <= bytes) return;
        if (p) { pool_free(p); p = nullptr; bytes = 0; }
        p = pool_alloc(n); bytes = n;
    }
    u64* u() const { return (u64*)p; }
    void release() { if (p) { pool_free(p); p = nullptr; bytes = 0; } }
};

// ---- run-time compiled kernels (expr_jit.hip): what the code-object cache did so far in this process ----
struct JitStats { uint64_t compiled = 0, disk_hits = 0, mem_hits = 0, spawned = 0; double ms = 0; };   // spawned: compilations done by a helper process (of `compiled`)
JitStats jit_stats();
// the next compilations of this thread go to a helper process each (a setup compiling several step programs at once: hipRTC is serial inside one process)
void jit_prefer_spawn(bool on);

// ---- NTT (ntt.hip) ----
// natural-order batched NTT over a row-major [1<<nbits][n_pols] device matrix; dst != src.
// `tmp` must hold (1<<nbits)*n_pols words when the plan has more than one pass.
void ntt_dev(const u64* d_src, u64* d_dst, u64* d_tmp, uint32_t n_pols, uint32_t nbits, bool inverse, hipStream_t st);
// low-degree extension on the coset 49*<w_ext>: [1<<nbits][n_pols] -> [1<<nbits_ext][n_pols].
// tmp: (1<<nbits_ext)*n_pols words.
void lde_dev(const u64* d_src, u64* d_dst, u64* d_tmp, uint32_t n_pols, uint32_t nbits, uint32_t nbits_ext, hipStream_t st);
int ntt_num_passes(uint32_t nbits);

// ---- Poseidon / Merkle (poseidon.hip) ----
void poseidon_dev(const u64* d_in8, const u64* d_cap4, u64* d_out, int n_out, hipStream_t st);
void linearhash_rows_dev(const u64* d_rows, uint32_t width, uint64_t height, u64* d_digests, hipStream_t st);
uint64_t merkle_n_nodes(uint64_t height);
// nodes: merkle_n_nodes(height)*4 words, zero-filled by this call; leaves from [height][width] rows
void merkelize_dev(const u64* d_rows, uint32_t width, uint64_t height, u64* d_nodes, hipStream_t st);
// n paths: from leaf digest [n][4] and siblings [n][max_depth][4] (path q uses its first depth[q] levels) to the root each implies
void merkle_roots_from_paths_dev(const u64* d_leaves, const u64* d_paths, const uint32_t* d_depth, const u64* d_idx, uint32_t n, uint32_t max_depth,
                                 u64* d_roots, hipStream_t st);

// ---- device-resident transcript (poseidon.hip; transcript.rs:8-103) ----
size_t transcript_state_bytes();
std::string poseidon_tables_selfcheck();   // host only: the one-lane kernels' matrix-pipe tables and arithmetic against 128-bit arithmetic ("" = fine)
void transcript_init_dev(void* d_t, hipStream_t st);
void transcript_put_dev(void* d_t, const u64* d_src, uint64_t n, hipStream_t st);
void transcript_get_dev(void* d_t, u64* d_dst, uint32_t n_words, hipStream_t st);
void transcript_permutations_dev(void* d_t, uint32_t n, uint32_t nbits, u64* d_dst, hipStream_t st);
void transcript_put_get_dev(void* d_t, const u64* d_src, uint64_t n_put, u64* d_dst, uint32_t n_get, uint32_t bits, hipStream_t st);

// ---- prover glue (stark.hip) ----
const u64* ntt_w256_table(bool inverse);  // w_256^(+-e), e < 256, device pointer (ntt.hip)
struct EvalDescHost { const u64* buf; uint64_t width; uint64_t offset; uint32_t dim; uint32_t prime; };
void fri_fold_dev(const u64* d_pol, uint32_t pol_bits, uint32_t step_bits, const u64* d_special_x, u64 shift_inv, u64* d_out, hipStream_t st);
void fri_transpose_dev(const u64* d_pol, uint64_t n, uint32_t tbits, u64* d_out, hipStream_t st);
void x_table_dev(uint32_t nbits, u64 shift, u64* d_out, hipStream_t st);
void zh_inv_dev(uint32_t nbits, uint32_t extend_bits, u64* d_out, hipStream_t st);
void xdivxsub_dev(const u64* d_xi, u64 mulw, uint32_t nbits_ext, u64* d_out, hipStream_t st);
struct EvalDescKHost { const u64* buf; const u64* L; uint64_t width; uint64_t offset; uint32_t dim; uint32_t rshift; };   // L: weights per row; rows k << rshift
void lev_dev(const u64* d_xi, uint32_t nbits, bool prime, u64 shift, u64* d_out, u64* d_pow, u64* d_tmp2, hipStream_t st);
void lev_pow_dev(const u64* d_xi, uint32_t nbits, bool prime, u64 shift, u64* d_pow, hipStream_t st);
void lev_pow_multi_dev(const u64* d_xi, uint32_t nbits, uint32_t n_tables, const bool* prime, const u64* shift, u64* const* d_pow, hipStream_t st);
void xdivxsub2_dev(const u64* d_xi, u64 mulw0, u64 mulw1, uint32_t nbits_ext, u64* d_out0, u64* d_out1, hipStream_t st);
void evals_k_dev(const EvalDescKHost* descs, uint32_t n_ev, uint32_t nbits, u64* d_out, hipStream_t st);
void evals_dev(const EvalDescHost* descs, uint32_t n_ev, uint32_t nbits, uint32_t ext, const u64* d_LEv, const u64* d_LpEv, u64* d_out, hipStream_t st);
void pol_get_dev(const u64* d_buf, uint64_t width, uint64_t offset, uint32_t dim, uint64_t n, u64* d_out, hipStream_t st);
void pol_set_dev(u64* d_buf, uint64_t width, uint64_t offset, uint32_t dim, uint64_t n, const u64* d_in, hipStream_t st);
// openings of a GL tree at n device-resident (already range-checked) indices, written to device memory as
// n x (width + 4 * depth) words: no host round trip (capi.hip; the prover batches the openings of all its trees)
void merkle_group_proofs_async(const struct ::zk_merkle* t, const u64* d_idx, uint32_t n, u64* d_out, hipStream_t st);
void merkle_group_proofs_multi_async(const struct ::zk_merkle* const* trees, const u64* masks, u64* const* d_outs, uint32_t n_trees, const u64* d_idx, uint32_t n, hipStream_t st);
uint32_t merkle_width(const struct ::zk_merkle* t);
uint64_t merkle_height(const struct ::zk_merkle* t);
void merkle_group_proofs_masked_async(const struct ::zk_merkle* t, const u64* d_idx, u64 mask, uint32_t n, u64* d_out, hipStream_t st);
void transcript_permutations_async(struct ::zk_transcript* t, uint32_t n, uint32_t nbits, u64* d_dst, hipStream_t st);
// put d_src[0..n_put) and squeeze n_get words (bits == 0) or n_get indices of `bits` bits, one launch on `st`
void transcript_put_get_async(struct ::zk_transcript* t, const u64* d_src, uint64_t n_put, u64* d_dst, uint32_t n_get, uint32_t bits, hipStream_t st);
uint64_t h1h2_work_words(uint64_t n);
void calculate_h1h2_dev(const u64* d_f, const u64* d_t, uint64_t n, u64* d_h1, u64* d_h2, u64* d_work, u64** d_missing, hipStream_t st);
void calculate_z_dev(const u64* d_num, const u64* d_den, uint64_t n, u64* d_z, u64* d_work, u64* d_check, hipStream_t st);
// ---- MSM (msm.hip): bases n*64 B affine Montgomery, scalars n*32 B canonical; d_out 17 u32 (x, y, inf flag)
void msm_g1_bn254_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st);
void g1_bn254_mul_generator_dev(const u64* d_k, uint64_t n, void* d_bases, hipStream_t st);  // P_i = [k_i]G, k_i != 0
// BLS12-381: bases n*96 B, scalars n*32 B; d_out 25 u32 (x, y, inf flag)
void msm_g1_bls12_381_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st);
void g1_bls12_381_mul_generator_dev(const u64* d_k, uint64_t n, void* d_bases, hipStream_t st);
// G2 (Fq2 coordinates: x.c0 || x.c1 || y.c0 || y.c1): 128 B / 192 B per point; d_out = that + a flag word
void msm_g2_bn254_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st);
void g2_bn254_mul_generator_dev(const u64* d_k, uint64_t n, void* d_bases, hipStream_t st);
void msm_g2_bls12_381_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st);
void g2_bls12_381_mul_generator_dev(const u64* d_k, uint64_t n, void* d_bases, hipStream_t st);
// window tables for fixed bases (msm_impl.hip.h): table[w * n + i] = 2^(16 w) P_i; a sum over points [off, off + n) of it
#define ZK_MSM_FIXED_DECL(NAME)                                                                                             \
    size_t msm_##NAME##_fixed_table_bytes(uint64_t n);                                                                      \
    void msm_##NAME##_fixed_prepare_dev(const void* d_bases, uint64_t n, void* d_table, hipStream_t st);                     \
    void msm_##NAME##_fixed_dev(const void* d_table, uint64_t table_n, uint64_t off, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st);
ZK_MSM_FIXED_DECL(g1_bn254)
ZK_MSM_FIXED_DECL(g2_bn254)
ZK_MSM_FIXED_DECL(g1_bls12_381)
ZK_MSM_FIXED_DECL(g2_bls12_381)
#undef ZK_MSM_FIXED_DECL
// ---- BN128-field hashing (poseidon_bn128.hip); digests = 4 raw (Montgomery, R = 2^256) limbs
void bn128_load_constants(const char* path);
std::string bn128_tables_selfcheck(const char* path);   // host only: the matrix-pipe tables (fr_mfma.hip.h) against the constants; "" or what is wrong
void bn128_poseidon_dev(const u64* d_inp, uint64_t n, uint32_t n_in, const u64* d_init, uint32_t n_out, u64* d_out, hipStream_t st);
uint64_t bn128_merkle_n_nodes(uint64_t height);
void bn128_linearhash_rows_dev(const u64* d_rows, uint32_t width, uint64_t height, u64* d_digests, hipStream_t st);
void bn128_merkelize_dev(const u64* d_rows, uint32_t width, uint64_t height, u64* d_nodes, hipStream_t st);
// the same over the BLS12-381 scalar field (verificationHashType "BLS12381")
void bls12381_load_constants(const char* path);
std::string bls12381_tables_selfcheck(const char* path);
void bls12381_poseidon_dev(const u64* d_inp, uint64_t n, uint32_t n_in, const u64* d_init, uint32_t n_out, u64* d_out, hipStream_t st);
uint64_t bls12381_merkle_n_nodes(uint64_t height);
void bls12381_linearhash_rows_dev(const u64* d_rows, uint32_t width, uint64_t height, u64* d_digests, hipStream_t st);
void bls12381_merkelize_dev(const u64* d_rows, uint32_t width, uint64_t height, u64* d_nodes, hipStream_t st);
// stark_verify (stark_verify.hip): 1 accepted, 0 rejected (`why` names the failed check); throws Error on malformed input
struct JVal;
int stark_verify_impl(const JVal& info, const JVal& prog, const JVal& ss, const u64 const_root[4], const char* zkin_json, std::string& why);
// the verifier's side of the scalar-field trees and digests (capi.hip)
bool fr_digest_from_dec(bool bls12381, const std::string& dec, u64 out[4]);
void fr_hash16_dev(bool bls12381, const u64* d_in, uint64_t n, const u64* d_zero4, u64* d_out, hipStream_t st);
void fr_linearhash_rows_dev(bool bls12381, const u64* d_rows, uint32_t width, uint64_t height, u64* d_digests, hipStream_t st);
void qsplit_dev(const u64* d_qq1, uint32_t nbits, uint32_t q_dim, uint32_t q_deg, u64* d_qq2, hipStream_t st);
// base-field elements in place: canonical integers <-> Montgomery (msm.hip); n = number of Fq elements
void fq_bn254_canon_to_mont_dev(void* d, uint64_t n, hipStream_t st);
void fq_bn254_mont_to_canon_dev(void* d, uint64_t n, hipStream_t st);
void fq_bls12_381_canon_to_mont_dev(void* d, uint64_t n, hipStream_t st);
void fq_bls12_381_mont_to_canon_dev(void* d, uint64_t n, hipStream_t st);

// ---- compressor12 exec (compressor12.hip): witness -> committed trace [n_rows][12]
struct C12Exec;
C12Exec* c12_exec_new(const char* exec_json, size_t len, uint64_t n_witness);
void c12_exec_free(C12Exec* e);
uint64_t c12_exec_levels(const C12Exec* e);
void c12_exec_dev(const C12Exec* e, const u64* d_witness, uint64_t n_witness, uint64_t n_rows, u64* d_cm, hipStream_t st);
// ---- Groth16 around the multi-scalar sums (groth16.hip): scalar-field transforms, the quotient, the prover ----
// bellman's EvaluationDomain::{fft, ifft, coset_fft, icoset_fft} on 2^logn Fr elements (4 x u64 Montgomery), in place
void fr_bn254_ntt_dev(u64* d_data, int logn, bool inverse, bool coset, hipStream_t st);
void fr_bls12_381_ntt_dev(u64* d_data, int logn, bool inverse, bool coset, hipStream_t st);
// a <- coefficients of (A B - C) / (X^n - 1) from the row evaluations a, b, c (prover.rs create_proof's h block)
void fr_bn254_quotient_dev(u64* d_a, const u64* d_b, const u64* d_c, int logn, hipStream_t st);
void fr_bls12_381_quotient_dev(u64* d_a, const u64* d_b, const u64* d_c, int logn, hipStream_t st);
namespace g16 {
struct Lc { std::vector<u32> col, coeff; };            // coeff: 8 x u32 canonical per term
struct Row { Lc lc[3]; };
struct R1cs { uint32_t n_wires = 0, n_pub_out = 0, n_pub_in = 0, n_prv_in = 0; std::vector<Row> rows; };
struct PointVec { uint64_t n = 0; std::vector<u32> w; std::vector<char> inf; };   // canonical little-endian words
struct Params { PointVec vk[6]; PointVec ic, h, l, a, b_g1, b_g2; };               // vk: alpha_g1 beta_g1 beta_g2 gamma_g2 delta_g1 delta_g2
std::string words_to_dec(const u32* w, int n);
}
struct Groth16Setup {
    virtual ~Groth16Setup() {}
    // witness: n_wires x 32 B canonical (host or device); proof_out: A || B || C affine Montgomery words; d_h_out:
    // optional device buffer for the quotient's (2^domain_log - 1) x 32 B canonical coefficients
    virtual void prove(const void* witness, bool on_device, const u64 r[4], const u64 s[4], u32* proof_out, std::string* json, u64* d_h_out) = 0;
    virtual uint32_t num_wires() const = 0;
    virtual uint32_t num_inputs() const = 0;
    virtual uint32_t domain_log() const = 0;
    std::string curve;
    std::vector<u32> modulus;     // Fr, 8 words
    size_t proof_words = 0;
};
Groth16Setup* groth16_setup_new(const char* curve, const void* r1cs, size_t r1cs_len, const void* params, size_t params_len);
void groth16_wtns_payload(const void* wtns, size_t len, const char* curve, uint64_t* offset, uint64_t* n);

}  // namespace zk
