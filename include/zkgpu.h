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
 *   - a process drives one GPU (hipSetDevice via zk_init); handles are not re-entrant.
 */
#ifndef ZKGPU_H
#define ZKGPU_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime ---------------------------------------------------------------------------- */
int zk_init(int device);            /* select the GPU this process proves on (default 0)      */
const char* zk_last_error(void);    /* message of the last failing call on this thread        */
int zk_device_count(void);          /* number of visible GPUs, <= 0 when none                  */
uint64_t zk_gl_modulus(void);       /* 0xFFFFFFFF00000001 (fields/src/field_gl.rs:12)          */
uint64_t zk_gl_root_of_unity(uint32_t k); /* MG.0[k] (starky/src/constant.rs:54-68), k <= 32   */

/* device memory helpers for callers that keep traces resident in HBM */
void* zk_dev_alloc(size_t bytes);
int zk_dev_free(void* d_ptr);
int zk_dev_upload(void* d_dst, const void* h_src, size_t bytes);
int zk_dev_download(void* h_dst, const void* d_src, size_t bytes);
int zk_dev_sync(void);

/* ---- NTT / LDE ---------------------------------------------------------------------------
 * replaces fft_p::fft / fft_p::ifft (starky/src/fft_p.rs:242-253):
 *   (buffsrc:&Vec<F>, n_pols, nbits, buffdst:&mut Vec<F>)  -- natural order in and out,
 *   forward root MG.0[nbits]; inverse includes the 1/N factor.  dst must not alias src.      */
int zk_gl_ntt(const uint64_t* src, uint64_t* dst, uint32_t n_pols, uint32_t nbits, int inverse);
/* replaces fft_p::interpolate (fft_p.rs:255-261): LDE on the coset 49*<w_ext>;
 * src [1<<nbits][n_pols] -> dst [1<<nbits_ext][n_pols]; n_pols == 0 is a no-op (:262-264).   */
int zk_gl_lde(const uint64_t* src, uint32_t n_pols, uint32_t nbits, uint64_t* dst, uint32_t nbits_ext);
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
uint32_t zk_merkle_depth(const zk_merkle_t* t);                     /* siblings per proof       */
/* get_group_proof(idx) (merklehash.rs:430-438): row_out[width], path_out[depth*4];
 * idx >= height is an error, as the reference bails.                                          */
int zk_merkle_group_proof(const zk_merkle_t* t, uint64_t idx, uint64_t* row_out, uint64_t* path_out);
const uint64_t* zk_merkle_elements_dev(const zk_merkle_t* t);       /* device pointer of rows   */
const uint64_t* zk_merkle_nodes_dev(const zk_merkle_t* t);          /* device pointer of nodes  */
int zk_merkle_free(zk_merkle_t* t);

#ifdef __cplusplus
}
#endif
#endif /* ZKGPU_H */
