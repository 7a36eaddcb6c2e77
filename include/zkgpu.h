/* libzkgpu -- C ABI of the MI355X (gfx950) prover backend for eigen-zkvm's starky hot path.
 *
 * Each entry point stands behind one Rust seam of the reference (there is no FFI on this path in
 * the reference; the seams are traits / free functions -- SURVEY.md section 8b).  The binding a
 * maintainer would add on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - field elements: canonical Goldilocks u64 (value < p = 2^64 - 2^32 + 1), little-endian, the
 *     same words the reference reads/writes in .cm/.const files (polsarray.rs:137-217) and
 *     obtains from Fr::as_int() (fields/src/field_gl.rs:542-544).
 *   - matrices: row-major [rows][cols], like the reference's buffers (polsarray.rs:219-227).
 *   - status: 0 = ok, nonzero = error; zk_last_error() returns the message for the calling
 *     thread (reference: anyhow::Result / panic).  Pointer-returning calls return NULL on error.
 *   - `*_dev` variants take DEVICE pointers (HBM-resident data) and a hipStream_t passed as
 *     void*; they enqueue work and return without synchronising.  The plain variants take HOST
 *     pointers, copy in/out and synchronise (drop-in for the reference's Vec<FGL> arguments).
 *   - a process drives one GPU, the one zk_init names: every host thread that calls into the library afterwards is
 *     bound to it on its first call (HIP's current device is per thread and a new thread starts on device 0).
 *     Without zk_init the threads stay on whatever device the caller gave them.  Handles are not re-entrant.
 */
#ifndef ZKGPU_H
#define ZKGPU_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime ---------------------------------------------------------------------------- */
int zk_init(int device);            /* select the GPU this process, all its threads, proves on  */
const char* zk_last_error(void);    /* message of the last failing call on this thread        */
int zk_device_count(void);          /* number of visible GPUs, <= 0 when none                  */
uint64_t zk_gl_modulus(void);       /* 0xFFFFFFFF00000001 (fields/src/field_gl.rs:12)          */
uint64_t zk_gl_root_of_unity(uint32_t k); /* MG.0[k] (starky/src/constant.rs:54-68), k <= 32   */

/* device memory helpers for callers that keep traces resident in HBM.  Blocks are cached by size
 * (a prover re-allocates the same multi-GB sections for every proof); zk_dev_trim returns the cache. */
void* zk_dev_alloc(size_t bytes);
int zk_dev_free(void* d_ptr);
int zk_dev_trim(void);
int zk_dev_upload(void* d_dst, const void* h_src, size_t bytes);
int zk_dev_download(void* h_dst, const void* d_src, size_t bytes);
int zk_dev_sync(void);
/* A non-blocking HIP stream of the library's own making (any hipStream_t of the caller's serves as well) -- for hosts
 * without a HIP binding of their own: one per concurrent prover thread.  zk_stream_free waits for the stream first;
 * call it from the thread that used the stream. */
void* zk_stream_new(void);
int zk_stream_sync(void* stream);
int zk_stream_free(void* stream);
int zk_dev_memset(void* d_ptr, int value, size_t bytes);
/* Synthetic field elements for benchmarks and size tests, generated where they are used: word i = splitmix64(seed + i), minus p
 * when that is >= p (the reference's benches fill their inputs with random field elements the same way, the files under starky/benches). */
int zk_dev_fill_splitmix(uint64_t* d, uint64_t n_words, uint64_t seed, void* stream);

/* ---- NTT / LDE ---------------------------------------------------------------------------
 * replaces fft_p::fft / fft_p::ifft (starky/src/fft_p.rs:242-253):
 *   (buffsrc:&Vec<F>, n_pols, nbits, buffdst:&mut Vec<F>)  -- natural order in and out,
 *   forward root MG.0[nbits]; inverse includes the 1/N factor.  dst must not alias src.      */
int zk_gl_ntt(const uint64_t* src, uint64_t* dst, uint32_t n_pols, uint32_t nbits, int inverse);
/* replaces fft_p::interpolate (fft_p.rs:255-261): LDE on the coset 49*<w_ext>;
 * src [1<<nbits][n_pols] -> dst [1<<nbits_ext][n_pols]; n_pols == 0 is a no-op (:262-264).   */
int zk_gl_lde(const uint64_t* src, uint32_t n_pols, uint32_t nbits, uint64_t* dst, uint32_t nbits_ext);
/* Streams.  Every `*_dev` entry point takes the HIP stream to issue on (NULL = the default stream).  Scratch memory
 * comes from a caching allocator inside the library; it orders the reuse of a block behind the work of every stream
 * the library has been handed so far, so calls on different streams (and from different host threads) may be mixed.
 * Handles (trees, transcripts, setups) are not re-entrant: one call at a time per handle.                            */
/* device-resident forms.  d_tmp: scratch of (1<<nbits)*n_pols words (ntt) or
 * (1<<nbits_ext)*n_pols words (lde); may be NULL when zk_gl_ntt_passes(nbits) == 1 (ntt only). */
int zk_gl_ntt_dev(const uint64_t* d_src, uint64_t* d_dst, uint64_t* d_tmp, uint32_t n_pols,
                  uint32_t nbits, int inverse, void* stream);
int zk_gl_lde_dev(const uint64_t* d_src, uint32_t n_pols, uint32_t nbits, uint64_t* d_dst,
                  uint64_t* d_tmp, uint32_t nbits_ext, void* stream);
int zk_gl_ntt_passes(uint32_t nbits); /* HBM passes a 2^nbits transform takes (bytes = 16*passes*N*n_pols) */

/* ---- Poseidon / LinearHash ---------------------------------------------------------------
 * replaces Poseidon::hash(inp, init_state, out) (starky/src/poseidon_opt.rs:76-200):
 * in[8] || cap[4] -> first n_out (1..12) state words.                                        */
int zk_gl_poseidon(const uint64_t in[8], const uint64_t cap[4], uint64_t* out, uint32_t n_out);
/* Host-only self check (NO GPU needed): the tables behind the kernels' matrix-pipe products -- the pre-sparse matrix P and the dense products
 * of the lazy partial-round blocks (poseidon_opt.rs:121-163) as balanced base-256 digit rows -- against their coefficients, and the
 * matrix-pipe arithmetic replayed step by step on the host against 128-bit arithmetic.  0 = fine, -1 = zk_last_error() says what is wrong. */
int zk_gl_poseidon_selfcheck(void);
/* replaces LinearHash::hash(flatvals, 0) (starky/src/linearhash.rs:79-110)                   */
int zk_gl_linearhash(const uint64_t* v, size_t n, uint64_t out[4]);
/* one digest per row of a device-resident [height][width] matrix -> d_digests[height][4]     */
int zk_gl_linearhash_rows_dev(const uint64_t* d_rows, uint32_t width, uint64_t height,
                              uint64_t* d_digests, void* stream);

/* ---- Merkle tree (trait MerkleTree, starky/src/traits.rs:24-55; MerkleTreeGL,
 *      starky/src/merklehash.rs:293-346, :430-457) ------------------------------------------ */
typedef struct zk_merkle zk_merkle_t;
uint64_t zk_merkle_n_nodes(uint64_t height);                       /* merklehash.rs:47-61      */
/* merkelize(buff, width, height): the tree copies `buff` to the device and owns it, as the
 * reference tree takes ownership of the Vec (merklehash.rs:326-329).                          */
zk_merkle_t* zk_gl_merkelize(const uint64_t* buff, uint32_t width, uint64_t height);
/* device-resident form: the tree BORROWS d_buff (must outlive the tree) and hashes on `stream` */
zk_merkle_t* zk_gl_merkelize_dev(const uint64_t* d_buff, uint32_t width, uint64_t height, void* stream);
int zk_merkle_root(const zk_merkle_t* t, uint64_t out[4]);          /* merklehash.rs:455-457    */
int zk_merkle_nodes(const zk_merkle_t* t, uint64_t* out);           /* all n_nodes*4 words      */
int zk_merkle_elements(const zk_merkle_t* t, uint64_t* out);        /* the committed rows, height*width words: what
                                                                       MerkleTree::to_extend reads back (merklehash.rs:260-265) */
uint32_t zk_merkle_depth(const zk_merkle_t* t);                     /* siblings per proof       */
/* get_group_proof(idx) (merklehash.rs:430-438): row_out[width], path_out[depth*4];
 * idx >= height is an error, as the reference bails.                                          */
int zk_merkle_group_proof(const zk_merkle_t* t, uint64_t idx, uint64_t* row_out, uint64_t* path_out);
/* n openings in one round trip (one launch, one copy): rows_out[n][width], paths_out[n][depth][4].  A proof opens every tree at
 * every query index (fri.rs:160-181); asked one by one, the launch + wait + copy of each opening is most of a small proof.    */
int zk_merkle_group_proofs(const zk_merkle_t* t, const uint64_t* idx, uint32_t n, uint64_t* rows_out, uint64_t* paths_out);
const uint64_t* zk_merkle_elements_dev(const zk_merkle_t* t);       /* device pointer of rows   */
const uint64_t* zk_merkle_nodes_dev(const zk_merkle_t* t);          /* device pointer of nodes  */
int zk_merkle_free(zk_merkle_t* t);

/* ---- Transcript (trait Transcript, starky/src/traits.rs:57-63; TranscriptGL,
 *      starky/src/transcript.rs:8-103).  The sponge state lives on the device: absorbing a root,
 *      the evals or the last FRI polynomial and drawing a challenge need no host round trip.     */
typedef struct zk_transcript zk_transcript_t;
zk_transcript_t* zk_transcript_new(void);                                        /* T::new()            */
int zk_transcript_put(zk_transcript_t* t, const uint64_t* src, size_t n);        /* put(), host words   */
int zk_transcript_put_dev(zk_transcript_t* t, const uint64_t* d_src, size_t n, void* stream);
int zk_transcript_get_field(zk_transcript_t* t, uint64_t out[3]);                /* get_field() -> host */
int zk_transcript_get_field_dev(zk_transcript_t* t, uint64_t* d_out3, void* stream);
int zk_transcript_get_fields1(zk_transcript_t* t, uint64_t* out);                /* get_fields1()       */
int zk_transcript_get_permutations(zk_transcript_t* t, uint32_t n, uint32_t nbits, uint64_t* out);
int zk_transcript_free(zk_transcript_t* t);

/* ---- FRI (starky/src/fri.rs:84-184) ----------------------------------------------------------
 * one folding step (fri.rs:101-126): d_pol [1<<pol_bits][3] -> d_out [1<<step_bits][3];
 * d_special_x = the step's challenge (3 device words); shift_inv = (49^-1)^(2^(nBitsExt-pol_bits)).
 * pol_bits - step_bits <= 11 (the reference folds any number of bits, fri.rs:112-126; its largest step is the 10 bits of
 * final.starkStruct.*.json, 2^17 -> 2^7).  step_bits == pol_bits copies (step 0 of the reference).                   */
int zk_fri_fold_dev(const uint64_t* d_pol, uint32_t pol_bits, uint32_t step_bits,
                    const uint64_t* d_special_x, uint64_t shift_inv, uint64_t* d_out, void* stream);
/* get_transposed_buffer (fri.rs:299-317): [n] F3G -> [1<<tbits][n>>tbits][3] words */
int zk_fri_transpose_dev(const uint64_t* d_pol, uint64_t n, uint32_t tbits, uint64_t* d_out, void* stream);

/* ---- stark_gen glue (starky/src/stark_gen.rs) --------------------------------------------------
 * x_n / x_2ns tables (:231-247): out[k] = shift * MG.0[nbits]^k                                    */
int zk_stark_x_table_dev(uint32_t nbits, uint64_t shift, uint64_t* d_out, void* stream);
/* build_Zh_Inv (:575-592): out[j] = 1/(49^(2^nbits) * MG.0[ext]^j - 1), j < 2^ext                 */
int zk_stark_zh_inv_dev(uint32_t nbits, uint32_t extend_bits, uint64_t* d_out, void* stream);
/* xDivXSubXi / xDivXSubWXi (:481-522): out[k] = x/(x - xi*mulw), x = 49*MG.0[nbits_ext]^k, [Next][3];
 * mulw = 1 for xi, MG.0[nBits] for w*xi                                                            */
int zk_stark_xdivxsub_dev(const uint64_t* d_xi, uint64_t mulw, uint32_t nbits_ext, uint64_t* d_out, void* stream);
/* LEv / LpEv (:416-430): iNTT_N of ((xi[*w])/49)^i -> [N][3]; d_tmp, d_tmp2: N*3 words each        */
int zk_stark_lev_dev(const uint64_t* d_xi, uint32_t nbits, int prime, uint64_t* d_out, uint64_t* d_tmp,
                     uint64_t* d_tmp2, void* stream);
/* evals (:432-466): out[e] = sum_k cell_e[(k<<ext)*width + offset] * L_e[k]; L_e = LpEv if prime   */
typedef struct { const uint64_t* d_buf; uint64_t width; uint64_t offset; uint32_t dim; uint32_t prime; } zk_eval_desc;
int zk_stark_evals_dev(const zk_eval_desc* descs, uint32_t n_ev, uint32_t nbits, uint32_t ext,
                       const uint64_t* d_LEv, const uint64_t* d_LpEv, uint64_t* d_out, void* stream);
/* Q split (:375-391): qq2[i][p*q_dim+k] = qq1[p*N+i][k] * (49^-N)^p ; qq2 is [Next][q_dim*q_deg]
 * and must be zero-filled by the caller beyond row N                                               */
int zk_stark_qsplit_dev(const uint64_t* d_qq1, uint32_t nbits, uint32_t q_dim, uint32_t q_deg,
                        uint64_t* d_qq2, void* stream);

/* get_pol / set_pol (:594-622, :683-707): column `offset` (1 or 3 words wide) of a [n][width] section
 * <-> a dense [n][3] polynomial (dim-1 columns are zero padded)                                    */
int zk_stark_get_pol_dev(const uint64_t* d_buf, uint64_t width, uint64_t offset, uint32_t dim, uint64_t n,
                         uint64_t* d_out3, void* stream);
int zk_stark_set_pol_dev(uint64_t* d_buf, uint64_t width, uint64_t offset, uint32_t dim, uint64_t n,
                         const uint64_t* d_in3, void* stream);
/* calculate_H1H2 (:624-651) over [n][3] operands: the sorted merge of the looked-up values f and the table t, split into its even
 * (h1) and odd (h2) entries.  A hash table over t on the device instead of the reference's HashMap + stable sort on the host;
 * synchronises and fails with "Number not included: <value>" for the first f the table lacks (:636-638). */
int zk_stark_calculate_h1h2_dev(const uint64_t* d_f3, const uint64_t* d_t3, uint64_t n, uint64_t* d_h1_3, uint64_t* d_h2_3, void* stream);
/* calculate_Z (:653-666): z[0] = 1, z[i] = z[i-1]*num[i-1]/den[i-1] over [n][3] operands; synchronises
 * and fails ("z does not close") when z[n-1]*num[n-1]/den[n-1] != 1, as the reference asserts (:663-664) */
int zk_stark_calculate_z_dev(const uint64_t* d_num3, const uint64_t* d_den3, uint64_t n, uint64_t* d_z3, void* stream);

/* ---- G1 multi-scalar multiplication (groth16 final wrap) ---------------------------------------
 * stands behind Groth16::prove (groth16/src/groth16.rs:88-96) -> bellman_ce::create_random_proof ->
 * multiexp, the seam bellperson occupies under `--features cuda` (groth16.rs:45-57).  Layout as
 * bellman keeps it: bases n x 64 B affine (x || y), Fq in Montgomery form (R = 2^256), little-endian
 * limbs; scalars n x 32 B canonical little-endian (FrRepr); out 64 B affine Montgomery, *is_infinity
 * set when the sum is the point at infinity.  `_dev` takes device pointers (d_out: 68 bytes) and, as
 * every `_dev` entry point, returns once the work is queued on `stream`.                              */
int zk_msm_g1_bn254(const void* bases, const void* scalars, uint64_t n, void* out, int* is_infinity);
int zk_msm_g1_bn254_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, void* stream);
/* the same on BLS12-381 G1 (zkit --curve BLS12381; pairing_ce bls12_381): Fq = 12 x 32-bit limbs, Montgomery
 * R = 2^384, bases n x 96 B, scalars n x 32 B canonical (255 bits), out 96 B; d_out of `_dev`: 100 bytes.  */
int zk_msm_g1_bls12_381(const void* bases, const void* scalars, uint64_t n, void* out, int* is_infinity);
int zk_msm_g1_bls12_381_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, void* stream);
int zk_g1_bls12_381_mul_generator_dev(const uint64_t* d_k, uint64_t n, void* d_bases, void* stream);
/* G2 (the B-query of a Groth16 proving key): the same entry points over the sextic twist, coordinates in
 * Fq2 = Fq[u]/(u^2 + 1).  A point is x.c0 || x.c1 || y.c0 || y.c1 (pairing_ce's G2Affine), each Fq as above:
 * 128 B (BN254) / 192 B (BLS12-381); `_dev` output = the point followed by a 4-byte infinity flag.              */
int zk_msm_g2_bn254(const void* bases, const void* scalars, uint64_t n, void* out, int* is_infinity);
int zk_msm_g2_bn254_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, void* stream);
int zk_g2_bn254_mul_generator_dev(const uint64_t* d_k, uint64_t n, void* d_bases, void* stream);
int zk_msm_g2_bls12_381(const void* bases, const void* scalars, uint64_t n, void* out, int* is_infinity);
int zk_msm_g2_bls12_381_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, void* stream);
int zk_g2_bls12_381_mul_generator_dev(const uint64_t* d_k, uint64_t n, void* d_bases, void* stream);
/* synthetic CRS for benches and tests: d_bases[i] = [d_k[i]] G, G = (1, 2), k_i a non-zero 64-bit integer
 * (what generate_random_parameters, groth16.rs:39,82, does with secret exponents).                     */
int zk_g1_bn254_mul_generator_dev(const uint64_t* d_k, uint64_t n, void* d_bases, void* stream);

/* ---- BN128-field hashing: verificationHashType "BN128", the final STARK of every aggregation --------------
 * (test/stark_aggregation.sh:199-210).  A digest is an ElementDigest<4, Fr> (starky/src/digest.rs:45-65): the
 * four RAW limbs of an Fr of BN254's scalar field, i.e. its Montgomery form a*2^256 mod r -- every uint64_t[4]
 * below carries that format, exactly what MTNodeType::from_scalar / as_scalar exchange.
 * zk_bn128_load_constants reads the Poseidon parameter tables (t = 2..17; data/poseidon_bn128_constants.bin,
 * written by tools/gen_poseidon_bn128_constants.py from poseidon_bn128_constants_opt.rs) once per device.      */
int zk_bn128_load_constants(const char* path);
/* Host-only check (no GPU) of the matrix-pipe tables the dense layers of the one-lane kernels run on (csrc/fr_mfma.hip.h): built from
 * the constants file, emulated the device's way on random and extreme vectors, compared with plain modular arithmetic.  0, or -1 with
 * zk_last_error() naming the first difference.                                                                                      */
int zk_bn128_poseidon_selfcheck(const char* path);
/* Poseidon::hash_ex(inp, init_state, out) (poseidon_bn128_opt.rs:80-86, 98-224): n_in = 1..16 inputs, t = n_in + 1,
 * first n_out (<= t) state words.  Wrong lengths are errors, as the reference bails (:99-105).                   */
int zk_bn128_poseidon(const uint64_t* inp, uint32_t n_in, const uint64_t init_state[4], uint32_t n_out, uint64_t* out);
/* batch form on device memory: inp [n][n_in][4], one shared init_state, out [n][n_out][4]                        */
int zk_bn128_poseidon_dev(const uint64_t* d_inp, uint64_t n, uint32_t n_in, const uint64_t* d_init_state, uint32_t n_out,
                          uint64_t* d_out, void* stream);
/* LinearHashBN128::hash_element_array (linearhash_bn128.rs:105-131): n Goldilocks words -> digest               */
int zk_bn128_linearhash(const uint64_t* v, size_t n, uint64_t out[4]);
/* MerkleTreeBN128 (merklehash_bn128.rs): arity 16; same ownership rules as zk_gl_merkelize(_dev)                */
typedef struct zk_bn128_merkle zk_bn128_merkle_t;
uint64_t zk_bn128_merkle_n_nodes(uint64_t height);                        /* :26-39                     */
zk_bn128_merkle_t* zk_bn128_merkelize(const uint64_t* buff, uint32_t width, uint64_t height);           /* :196-239 */
zk_bn128_merkle_t* zk_bn128_merkelize_dev(const uint64_t* d_buff, uint32_t width, uint64_t height, void* stream);
int zk_bn128_merkle_root(const zk_bn128_merkle_t* t, uint64_t out[4]);   /* :275-277                    */
int zk_bn128_merkle_nodes(const zk_bn128_merkle_t* t, uint64_t* out);    /* all n_nodes*4 words         */
uint32_t zk_bn128_merkle_depth(const zk_bn128_merkle_t* t);
/* get_group_proof (:246-254): row_out[width], path_out[depth][16][4]; idx >= height is an error          */
int zk_bn128_merkle_group_proof(const zk_bn128_merkle_t* t, uint64_t idx, uint64_t* row_out, uint64_t* path_out);
/* n openings at once: rows_out[n][width], paths_out[n][depth][16][4] */
int zk_bn128_merkle_group_proofs(const zk_bn128_merkle_t* t, const uint64_t* idx, uint32_t n, uint64_t* rows_out, uint64_t* paths_out);
int zk_bn128_merkle_free(zk_bn128_merkle_t* t);
/* TranscriptBN128 (transcript_bn128.rs:14-132): put takes one Goldilocks word (n = 1) or one digest (n = 4)       */
typedef struct zk_bn128_transcript zk_bn128_transcript_t;
zk_bn128_transcript_t* zk_bn128_transcript_new(void);
int zk_bn128_transcript_put(zk_bn128_transcript_t* t, const uint64_t* e, size_t n);
int zk_bn128_transcript_get_fields1(zk_bn128_transcript_t* t, uint64_t* out);
int zk_bn128_transcript_get_field(zk_bn128_transcript_t* t, uint64_t out[3]);
int zk_bn128_transcript_get_permutations(zk_bn128_transcript_t* t, uint32_t n, uint32_t nbits, uint64_t* out);
int zk_bn128_transcript_free(zk_bn128_transcript_t* t);

/* ---- the same over the BLS12-381 scalar field: verificationHashType "BLS12381" (`--curve BLS12381`).  Twins of the
 * BN128 entry points above, behind poseidon_bls12381_opt.rs (hash() returns state[1], :94-103), linearhash_bls12381.rs,
 * merklehash_bls12381.rs, transcript_bls12381.rs; tables in data/poseidon_bls12381_constants.bin.                     */
typedef struct zk_bls12381_merkle zk_bls12381_merkle_t;
typedef struct zk_bls12381_transcript zk_bls12381_transcript_t;
int zk_bls12381_load_constants(const char* path);
int zk_bls12381_poseidon_selfcheck(const char* path);
int zk_bls12381_poseidon(const uint64_t* inp, uint32_t n_in, const uint64_t init_state[4], uint32_t n_out, uint64_t* out);
int zk_bls12381_poseidon_dev(const uint64_t* d_inp, uint64_t n, uint32_t n_in, const uint64_t* d_init_state, uint32_t n_out,
                          uint64_t* d_out, void* stream);
int zk_bls12381_linearhash(const uint64_t* v, size_t n, uint64_t out[4]);
uint64_t zk_bls12381_merkle_n_nodes(uint64_t height);
zk_bls12381_merkle_t* zk_bls12381_merkelize(const uint64_t* buff, uint32_t width, uint64_t height);
zk_bls12381_merkle_t* zk_bls12381_merkelize_dev(const uint64_t* d_buff, uint32_t width, uint64_t height, void* stream);
int zk_bls12381_merkle_root(const zk_bls12381_merkle_t* t, uint64_t out[4]);
int zk_bls12381_merkle_nodes(const zk_bls12381_merkle_t* t, uint64_t* out);
uint32_t zk_bls12381_merkle_depth(const zk_bls12381_merkle_t* t);
int zk_bls12381_merkle_group_proof(const zk_bls12381_merkle_t* t, uint64_t idx, uint64_t* row_out, uint64_t* path_out);
int zk_bls12381_merkle_group_proofs(const zk_bls12381_merkle_t* t, const uint64_t* idx, uint32_t n, uint64_t* rows_out, uint64_t* paths_out);
int zk_bls12381_merkle_free(zk_bls12381_merkle_t* t);
zk_bls12381_transcript_t* zk_bls12381_transcript_new(void);
int zk_bls12381_transcript_put(zk_bls12381_transcript_t* t, const uint64_t* e, size_t n);
int zk_bls12381_transcript_get_fields1(zk_bls12381_transcript_t* t, uint64_t* out);
int zk_bls12381_transcript_get_field(zk_bls12381_transcript_t* t, uint64_t out[3]);
int zk_bls12381_transcript_get_permutations(zk_bls12381_transcript_t* t, uint32_t n, uint32_t nbits, uint64_t* out);
int zk_bls12381_transcript_free(zk_bls12381_transcript_t* t);


/* ---- whole prover (starky/src/prove.rs:95-160: StarkSetup::new + StarkProof::stark_gen + FRI::prove) ------
 * zk_stark_setup_new stands behind StarkSetup::new (stark_setup.rs:27-66): it takes the reference's
 * serialised code-generator output, {"starkinfo": StarkInfo, "program": Program} (serde field names,
 * starkinfo.rs:27-95), the StarkStruct JSON (verificationHashType "GL", or "BN128" after
 * zk_bn128_load_constants: MerkleTreeBN128 + TranscriptBN128, prove.rs:47-61) and the constant
 * polynomials ([2^nBits][n_constants] words, the .const file, polsarray.rs:137-217); it extends and
 * merkelizes the constants on the device and compiles the step programs for gfx950.
 * zk_stark_gen stands behind StarkProof::stark_gen (stark_gen.rs:193-202) + FRI::prove (fri.rs:84-89):
 * cm_pols = the committed trace ([2^nBits][n_cm1] words, the .cm file).  Returns the proof as the
 * zkin JSON the reference's serialiser writes (serializer.rs:146-261), malloc'ed: release it with
 * zk_string_free.  NULL on error (zk_last_error), e.g. "z does not close" (stark_gen.rs:663-664).
 * A setup is bound to the device current at creation and is not re-entrant.                         */
/* The code generator: stands behind StarkInfo::new (starky/src/starkinfo.rs:160-272; called from StarkSetup::new,
 * stark_setup.rs:45-52).  pil_json = the compiled PIL (types.rs:134-155, what pilcom writes), stark_struct_json = the
 * StarkStruct.  Returns the JSON zk_stark_setup_new takes, {"starkinfo": StarkInfo, "program": Program} in the reference's
 * serde names, malloc'ed (zk_string_free); NULL on error, e.g. "stark_deg != pil_deg", "Global.L1 must be defined".
 * Host only.  A caller that has the Rust front end passes its own serde output instead and never needs this. */
char* zk_starkinfo_generate(const char* pil_json, const char* stark_struct_json);
typedef struct zk_stark_setup zk_stark_setup_t;
zk_stark_setup_t* zk_stark_setup_new(const char* starkinfo_program_json, const char* stark_struct_json,
                                     const uint64_t* const_pols, uint64_t n_words);
int zk_stark_setup_const_root(const zk_stark_setup_t* s, uint64_t out[4]);   /* StarkSetup.const_root */
/* stark_gen's `prover_addr` argument (stark_gen.rs:201): echoed as "proverAddr" by non-GL proofs (serializer.rs:255-262) */
int zk_stark_setup_set_prover_addr(zk_stark_setup_t* s, const char* prover_addr);
/* stark_verify (starky/src/stark_verify.rs:20-136, fri.rs:187-297) on a proof in the prover's own output format, the zkin
 * JSON of serializer.rs:146-261, for all three hash types: 1 = accepted, 0 = rejected (zk_last_error() names the failed
 * check: "Q != C * P", "FRIVerifierFailed: ...", a fold mismatch, the last polynomial's degree), -1 = malformed input.
 * The sponge, the LinearHash of every opened row and every Merkle path run in the library's kernels; the two short
 * verifier programs and the FRI groups' inverse transforms are scalar host work.  This is a verifier for UNTRUSTED zkin, stricter
 * than the reference where the reference is loose: scalar-field (16-ary) Merkle paths are walked level by level -- the node at position
 * idx & 15 of every level must be the value carried up, starting from the row's digest -- whereas merklehash_bn128.rs:108-128 binds only
 * the last level to the root (rows unbound); finalPol must hold exactly 2^steps.last values and every path must have the depth its tree
 * implies.  Honest proofs pass either way.  zk_stark_verify_set_reference_compat(1) switches the CALLING THREAD to the reference's lenient path
 * check (for parity tests against a verifier that follows the reference to the letter); it returns the previous setting.
 *   zk_stark_verify       against a prover's setup (its StarkInfo / Program / StarkStruct and the root of its constants)
 *   zk_stark_verify_with  without one: the same JSON texts zk_stark_setup_new takes + const_root (GL words, or the raw
 *                         Montgomery limbs zk_stark_setup_const_root returns for BN128 / BLS12381)
 * zk_stark_setup_set_self_check(s, 1): every later zk_stark_gen* of the setup verifies its proof before returning it and
 * fails (NULL, "the proof does not verify: ...") otherwise -- the `assert!(stark_verify(..))` of prove.rs:124-132.        */
int zk_stark_verify(const zk_stark_setup_t* s, const char* zkin_json);
int zk_stark_verify_with(const char* starkinfo_program_json, const char* stark_struct_json, const uint64_t const_root[4],
                         const char* zkin_json);
int zk_stark_verify_set_reference_compat(int on);
int zk_stark_setup_set_self_check(zk_stark_setup_t* s, int on);
/* Where the time went, as JSON text owned by the setup (valid until the next call on it / its release).
 * zk_stark_setup_timing: StarkSetup::new (stark_setup.rs:26-66, one `#[time_profiler("stark_setup")]` span there) split into
 *   json_parse_ms, const_lde_merkle_ms, programs_ms (+ how many step programs hipRTC compiled and how many came from the
 *   code-object cache: $ZK_JIT_CACHE, default ~/.cache/zkgpu, "off" disables; `hiprtc_processes`: how many of the compilations ran
 *   in a helper process -- a setup compiles its step programs side by side, one `zkgpu_jitc` (next to libzkgpu.so; $ZK_JITC names
 *   another, "off" disables) per program, because hipRTC compiles serially inside one process; a code object is sha256-checked when it
 *   comes back from disk, and a cache directory that is not the user's own or is writable by others is not used).
 * zk_stark_last_timing: the stages of the last proof of this setup in HIP-event milliseconds, named after the reference's
 *   spans (stark_gen.rs:192,624,709,734,785; fri.rs:83): extend, merkelize, calculate_exps_parallel, calculate_H1H2,
 *   calculate_Z, fri_prove, ...  Collected only when the environment has ZK_STARK_TIMING=1 (also logged to stderr); "" otherwise. */
const char* zk_stark_setup_timing(const zk_stark_setup_t* s);
const char* zk_stark_last_timing(const zk_stark_setup_t* s);
char* zk_stark_gen(zk_stark_setup_t* s, const uint64_t* cm_pols, uint64_t n_words);
/* same with the trace already resident in HBM (borrowed, not modified), e.g. written there by a device-side
 * witness generator or uploaded while the previous proof was running                                     */
char* zk_stark_gen_dev(zk_stark_setup_t* s, const uint64_t* d_cm_pols, uint64_t n_words);
/* The same on a stream of the caller's.  Proofs of DIFFERENT setups may run at the same time from different host threads,
 * each on its own (non-blocking) stream: small proofs are bound by the latency of their launch chain, not by the device
 * (test/stark_aggregation.sh:70-73 runs its recursion tasks as parallel processes for the same reason).  One setup serves
 * one proof at a time. */
char* zk_stark_gen_dev_on(zk_stark_setup_t* s, const uint64_t* d_cm_pols, uint64_t n_words, void* stream);

/* ---- the staged prover: stark_gen cut at the reference's own seams ----
 * For a caller that keeps its own stark_gen.rs and swaps in the device for the heavy calls it makes:
 *   calculate_exps_parallel(ctx, starkinfo, segment, domain, step)   stark_gen.rs:786-792   -> zk_stark_eval
 *   extend_and_merkelize(ctx, stage) + transcript.put(root)          stark_gen.rs:709-750   -> zk_stark_commit_stage
 *   transcript.get_field() into ctx.challenges[i]                    stark_gen.rs:285-294   -> zk_stark_challenge (or zk_stark_set_challenge)
 *   calculate_H1H2 / calculate_Z over every argument of the PIL      stark_gen.rs:300-353   -> zk_stark_calculate_h1h2 / _z
 *   the evaluations at xi, w xi + their absorption                   stark_gen.rs:416-472   -> zk_stark_evals
 *   FRI::prove(transcript, pol, query_pol)                           fri.rs:84-184          -> zk_stark_fri_prove, or zk_fri_prove_dev alone
 * A context (one proof in progress) belongs to one setup and one stream; its sections stay in HBM between the calls.  The calls are
 * accepted in the reference's order only (an out-of-order call fails with a message naming what comes first):
 *   new -> commit 1 -> challenge 0, 1 -> eval 2PREV -> calculate_h1h2 -> commit 2 -> challenge 2, 3 -> eval 3PREV -> calculate_z ->
 *   eval 3 -> commit 3 -> challenge 4 -> eval 42NS -> commit 4 -> challenge 7 -> evals -> challenge 5, 6 -> eval 52NS -> fri_prove -> finish.
 * zk_stark_gen* is exactly this sequence; a proof driven through the stages is byte-equal to zk_stark_gen's (tests/cabi/zkgpu_cabi_test.c).
 *   zk_stark_new            the trace in host memory (cm_pols) or in HBM (d_cm_pols), the other NULL; sections allocated, publics computed and absorbed.
 *                           d_cm_pols is BORROWED for the life of the context: every later stage reads it; it must stay allocated and unchanged
 *                           until zk_stark_free (a host trace is copied, the caller may release it when zk_stark_new returns)
 *   zk_stark_commit_stage   stage 1..3: LDE + Merkle tree of cm<stage>; stage 4: Q split (inverse NTT, split, NTT) + tree 4; the root is absorbed by the
 *                           context's transcript and copied to root[4] when non-NULL (GL words, or the raw Montgomery limbs of a scalar-field digest)
 *   zk_stark_challenge      challenge i <- the context's transcript (copied to out[3] when non-NULL); i: 0 u, 1 defVal, 2 gamma, 3 beta, 4 vc, 5 v1, 6 v2, 7 xi
 *   zk_stark_set_challenge  challenge i <- v: for a caller whose own sponge does Fiat-Shamir; such a caller also runs FRI through zk_fri_prove_dev
 *                           with its own transcript (zk_stark_fri_pol_dev, zk_stark_tree give it the polynomial and the query trees)
 *   zk_stark_evals          -> number of evaluations; 3 words each into evals_out (host, may be NULL)
 *   zk_stark_finish         the openings, the one read-back, the zkin JSON (malloc'ed: zk_string_free); honours zk_stark_setup_set_self_check
 *   zk_fri_prove_dev        FRI::prove on any device polynomial of 2^nbits_ext extension values (3 words each) with the caller's TranscriptGL and the
 *                           GL trees its queries open: -> JSON {"ys", "s<k>_root", "s<k>_vals", "s<k>_siblings", "s0_vals<j>", "s0_siblings<j>", "finalPol"}
 *                           (query trees numbered from 1; serializer.rs:189-252's layout), malloc'ed                                              */
typedef struct zk_stark_ctx zk_stark_ctx_t;
enum { ZK_STEP_2PREV = 0, ZK_STEP_3PREV = 1, ZK_STEP_3 = 2, ZK_STEP_42NS = 3, ZK_STEP_52NS = 4 };
zk_stark_ctx_t* zk_stark_new(zk_stark_setup_t* s, const uint64_t* cm_pols, const uint64_t* d_cm_pols, uint64_t n_words, void* stream);
int zk_stark_commit_stage(zk_stark_ctx_t* c, int stage, uint64_t root[4]);
int zk_stark_challenge(zk_stark_ctx_t* c, int i, uint64_t out[3]);
int zk_stark_set_challenge(zk_stark_ctx_t* c, int i, const uint64_t v[3]);
int zk_stark_eval(zk_stark_ctx_t* c, int step);
int zk_stark_calculate_h1h2(zk_stark_ctx_t* c);
int zk_stark_calculate_z(zk_stark_ctx_t* c);
int zk_stark_evals(zk_stark_ctx_t* c, uint64_t* evals_out, uint64_t cap_words);
int zk_stark_fri_prove(zk_stark_ctx_t* c);
char* zk_stark_finish(zk_stark_ctx_t* c);
const uint64_t* zk_stark_fri_pol_dev(const zk_stark_ctx_t* c);            /* f_2ns after step 52NS: FRI's input, 3 << nBitsExt words in HBM */
const zk_merkle_t* zk_stark_tree(const zk_stark_ctx_t* c, int j);         /* 1..4 once committed, 5 = the constants' tree; NULL for scalar-field hashing */
int zk_stark_free(zk_stark_ctx_t* c);
char* zk_fri_prove_dev(zk_transcript_t* transcript, const uint64_t* d_pol, uint32_t nbits_ext, const uint32_t* steps, uint32_t n_steps,
                       uint32_t n_queries, const zk_merkle_t* const* query_trees, uint32_t n_query_trees, void* stream);
void zk_string_free(char* s);
int zk_stark_setup_free(zk_stark_setup_t* s);

/* Window tables for bases that do not change between calls (a Groth16 proving key): the table holds 2^(16 w) P_i for the
 * 16 windows of every base (16 x the size of the internal point form; *_table_bytes says how much), built once on the
 * device; a sum over the points [offset, offset + n) of a table built for table_n bases then needs no doublings.
 * table_n < 2^24.  Same operand formats and the same (bit-identical) result as zk_msm_*_dev.                        */
size_t zk_msm_g1_bn254_table_bytes(uint64_t table_n);
int zk_msm_g1_bn254_table_build_dev(const void* d_bases, uint64_t table_n, void* d_table, void* stream);
int zk_msm_g1_bn254_table_dev(const void* d_table, uint64_t table_n, uint64_t offset, const void* d_scalars, uint64_t n, void* d_out, void* stream);
size_t zk_msm_g2_bn254_table_bytes(uint64_t table_n);
int zk_msm_g2_bn254_table_build_dev(const void* d_bases, uint64_t table_n, void* d_table, void* stream);
int zk_msm_g2_bn254_table_dev(const void* d_table, uint64_t table_n, uint64_t offset, const void* d_scalars, uint64_t n, void* d_out, void* stream);
size_t zk_msm_g1_bls12_381_table_bytes(uint64_t table_n);
int zk_msm_g1_bls12_381_table_build_dev(const void* d_bases, uint64_t table_n, void* d_table, void* stream);
int zk_msm_g1_bls12_381_table_dev(const void* d_table, uint64_t table_n, uint64_t offset, const void* d_scalars, uint64_t n, void* d_out, void* stream);
size_t zk_msm_g2_bls12_381_table_bytes(uint64_t table_n);
int zk_msm_g2_bls12_381_table_build_dev(const void* d_bases, uint64_t table_n, void* d_table, void* stream);
int zk_msm_g2_bls12_381_table_dev(const void* d_table, uint64_t table_n, uint64_t offset, const void* d_scalars, uint64_t n, void* d_out, void* stream);

/* ---- Groth16 around the multi-scalar sums (`zkit groth16_prove`, groth16/src/api.rs:144-205) ----
 * The reference hands the whole proof to bellman_ce::groth16::create_random_proof (groth16/src/groth16.rs:88-96;
 * third-party).  These entry points keep everything after witness generation on the device.
 *
 * zk_fr_*_ntt stand behind bellman's EvaluationDomain::{fft, ifft, coset_fft, icoset_fft}: 2^log_n elements of
 * the curve's scalar field (4 x u64 Montgomery R = 2^256, i.e. the raw limbs of an Fr), natural order in and
 * out, in place; omega = Fr::root_of_unity()^(2^(S - log_n)), coset generator Fr::multiplicative_generator() = 7.
 * zk_fr_*_quotient stand behind create_proof's `h` block: from the per-row evaluations a, b, c (2^log_n each)
 * to the coefficients of (A B - C)/(X^n - 1), written over a (bellman then drops the last one).              */
int zk_fr_bn254_ntt(uint64_t* data, uint32_t log_n, int inverse, int coset);
int zk_fr_bn254_ntt_dev(uint64_t* d_data, uint32_t log_n, int inverse, int coset, void* stream);
int zk_fr_bls12_381_ntt(uint64_t* data, uint32_t log_n, int inverse, int coset);
int zk_fr_bls12_381_ntt_dev(uint64_t* d_data, uint32_t log_n, int inverse, int coset, void* stream);
int zk_fr_bn254_quotient_dev(uint64_t* d_a, const uint64_t* d_b, const uint64_t* d_c, uint32_t log_n, void* stream);
int zk_fr_bls12_381_quotient_dev(uint64_t* d_a, const uint64_t* d_b, const uint64_t* d_c, uint32_t log_n, void* stream);
/* zk_groth16_setup_new stands behind api.rs:161-171 (read_pk_from_file + create_circuit_from_file): `curve` is the
 * reference's curve_type string ("BN128" | "BLS12381"), `r1cs` the bytes of the circom .r1cs file
 * (algebraic/src/r1cs_file.rs), `params` the bytes of the proving key as bellman's Parameters::write lays it out
 * (unchecked read: points are not tested for curve membership, as with checked = false).  The circuit is the one
 * algebraic/src/circom_circuit.rs:94-160 synthesises; a key whose query sizes do not match it is an error.
 * zk_groth16_prove: `witness` = n_wires x 32 B little-endian canonical values (the body of a .wtns file, see
 * zk_groth16_wtns_payload; reader.rs:86-137), r and s = the two blinding scalars create_random_proof draws from its
 * rng (canonical, < the field modulus), `proof` (optional) receives A || B || C as affine Montgomery coordinates
 * (BN254: 64 + 128 + 64 B; BLS12-381: 96 + 192 + 96 B; G2 as x.c0 || x.c1 || y.c0 || y.c1).  Returns proof.json
 * as json_utils.rs:305-315 renders it (malloc'ed; zk_string_free), NULL on error.  A setup is bound to the device
 * current at its creation, keeps the key resident there (as window tables, 16 x the key, when it has fewer than 2^24
 * G1 bases) and is not re-entrant: one zk_groth16_prove at a time per setup.                                      */
/* base-field elements in place on the device: to_mont = 1: canonical integers (what key files and proof.json carry,
 * Fq::from_repr) -> Montgomery limbs (what the multi-scalar sums take); to_mont = 0: the inverse (Fq::into_repr).
 * n_elems elements of 32 B (BN254) / 48 B (BLS12-381), little-endian words.                                    */
int zk_fq_bn254_convert_dev(void* d_elems, uint64_t n_elems, int to_mont, void* stream);
int zk_fq_bls12_381_convert_dev(void* d_elems, uint64_t n_elems, int to_mont, void* stream);
typedef struct zk_groth16_setup zk_groth16_setup_t;
zk_groth16_setup_t* zk_groth16_setup_new(const char* curve, const void* r1cs, size_t r1cs_len, const void* params, size_t params_len);
int zk_groth16_setup_info(const zk_groth16_setup_t* s, uint32_t* n_wires, uint32_t* n_inputs, uint32_t* domain_log);
char* zk_groth16_prove(zk_groth16_setup_t* s, const void* witness, uint64_t n_wires, const uint64_t r[4], const uint64_t s_[4], void* proof);
/* same with the witness already in HBM (its values are taken as canonical: the host-side range check of zk_groth16_prove is
 * not repeated); d_h (optional) receives the quotient's 2^domain_log - 1 canonical coefficients */
char* zk_groth16_prove_dev(zk_groth16_setup_t* s, const void* d_witness, uint64_t n_wires, const uint64_t r[4], const uint64_t s_[4], void* proof, uint64_t* d_h);
int zk_groth16_wtns_payload(const void* wtns, size_t len, const char* curve, uint64_t* offset, uint64_t* n_values);
int zk_groth16_setup_free(zk_groth16_setup_t* s);

/* ---- compressor12 exec (recursion/src/compressor12/compressor12_exec.rs:17-103) ----------------
 * The step between a recursive circuit's circom witness and the committed trace of its STARK: the PlonkAdd sums
 * appended to the witness (:60-66) and the s_map gather into the 12 columns of Compressor.a (:72-94).
 * zk_c12_exec_new takes the text of the .exec file (a JSON array of u64, compressor12_setup.rs:51-83) and the length of
 * the circuit's witness; zk_c12_exec_dev writes the [n_rows][12] matrix (the .cm file's content; rows beyond
 * s_map_column_len are zero) from a witness of one u64 per wire, both in HBM -- ready for zk_stark_gen_dev.            */
typedef struct zk_c12_exec zk_c12_exec_t;
zk_c12_exec_t* zk_c12_exec_new(const char* exec_json, size_t len, uint64_t n_witness);
int zk_c12_exec_dev(const zk_c12_exec_t* e, const uint64_t* d_witness, uint64_t n_witness, uint64_t n_rows, uint64_t* d_cm, void* stream);
uint64_t zk_c12_exec_depth(const zk_c12_exec_t* e);   /* launches of the addition phase (longest chain of dependent sums) */
int zk_c12_exec_free(zk_c12_exec_t* e);

/* ---- constraint evaluation (starky/src/interpreter.rs:91-225, stark_gen.rs:752-963) ------------
 * A step's program is the reference's Segment.first (Vec<Section{op,dest,src}>,
 * starkinfo_codegen.rs:76-89) with every Node resolved to an address exactly as
 * interpreter.rs get_ref/set_ref/eval_map do (:286-524): section cells become
 * (buffer slot, column offset, row stride, prime); tmp/number/public/challenge/eval/x/Zi/xDivXSub*
 * keep their meaning.  zk_program_compile translates it to a gfx950 kernel (hipRTC) once;
 * zk_program_run_dev evaluates it for every row i of the domain, reading rows (i + next*prime) % N.
 * Value widths (the F3G dim) are inferred from the operands: dim-1 op dim-1 -> 1, anything with a
 * dim-3 operand -> 3 (f3g.rs:323-449); a dim-3 result written to a section cell occupies 3 words,
 * a dim-1 result 1 word (interpreter.rs:149-159).                                                */
enum { ZK_OP_ADD = 0, ZK_OP_SUB = 1, ZK_OP_MUL = 2, ZK_OP_COPY = 3 };
enum { ZK_OPND_TMP = 0, ZK_OPND_MEM = 1, ZK_OPND_NUMBER = 2, ZK_OPND_PUBLIC = 3, ZK_OPND_CHALLENGE = 4,
       ZK_OPND_EVAL = 5, ZK_OPND_X = 6, ZK_OPND_ZI = 7, ZK_OPND_XDIVXSUBXI = 8, ZK_OPND_XDIVXSUBWXI = 9 };
typedef struct {
    uint8_t kind;     /* ZK_OPND_*                                                            */
    uint8_t dim;      /* MEM: width of the cell (1 or 3); ignored for the other kinds          */
    uint8_t prime;    /* MEM: read row (i + next) % N                                          */
    uint8_t buf;      /* MEM: slot in zk_eval_ctx.bufs                                         */
    uint32_t id;      /* TMP: id; MEM: column offset; PUBLIC/CHALLENGE/EVAL: index             */
    uint32_t stride;  /* MEM: words per row of the section                                     */
    uint32_t _pad;
    uint64_t value;   /* NUMBER: canonical value                                               */
} zk_operand;
typedef struct { uint32_t op; uint32_t _pad; zk_operand dest; zk_operand src[2]; } zk_instr;
typedef struct {
    uint64_t* bufs[16];            /* device base pointers of the sections the program touches  */
    const uint64_t* publics;       /* [n_publics]                                               */
    const uint64_t* challenges;    /* [8][3]  (constant.rs:39-50)                               */
    const uint64_t* evals;         /* [n_evals][3]                                              */
    const uint64_t* x;             /* x_n or x_2ns, [N]                                         */
    const uint64_t* zi;            /* ZHInv table, [zi_mask + 1]                                */
    uint64_t zi_mask;
    const uint64_t* xdivxsubxi;    /* [N][3]                                                    */
    const uint64_t* xdivxsubwxi;   /* [N][3]                                                    */
} zk_eval_ctx;
typedef struct zk_program zk_program_t;
zk_program_t* zk_program_compile(const zk_instr* code, uint32_t n_instr);    /* needs no GPU */
const char* zk_program_source(const zk_program_t* p);                         /* generated HIP text */
/* what the code-object cache of the step kernels did so far in this process: out = {compiled by hipRTC, taken from
 * $ZK_JIT_CACHE (default ~/.cache/zkgpu; "off" disables), taken from memory} */
void zk_jit_cache_stats(uint64_t out[3]);
int zk_program_run_dev(zk_program_t* p, const zk_eval_ctx* ctx, uint32_t nbits_domain, uint64_t next, void* stream);
/* The same for rows row0 .. row0 + count - 1 only (the domain is still 2^nbits_domain rows: primed reads wrap around it).
 * calculate_exp_at_point (stark_gen.rs:558-572) is this with count = 1: a public calculator is evaluated at its one row. */
int zk_program_run_rows_dev(zk_program_t* p, const zk_eval_ctx* ctx, uint32_t nbits_domain, uint64_t next, uint64_t row0, uint64_t count,
                            void* stream);
int zk_program_free(zk_program_t* p);

#ifdef __cplusplus
}
#endif
#endif /* ZKGPU_H */
