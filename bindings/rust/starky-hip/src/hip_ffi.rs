//! Raw declarations of include/zkgpu.h (the subset the seams below use).  Field elements cross the boundary as canonical
//! `u64` words -- `FGL::as_int()` on the way in, `FGL::from(u64)` on the way out (fields/src/field_gl.rs:542, :328).
#![allow(non_camel_case_types)]
use std::ffi::{c_char, c_int, c_void, CStr};

#[repr(C)]
pub struct zk_merkle_t {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zk_transcript_t {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zk_stark_setup_t {
    _private: [u8; 0],
}
/// one proof in progress (the staged prover: stark_gen cut at its own seams)
#[repr(C)]
pub struct zk_stark_ctx_t {
    _private: [u8; 0],
}
pub const ZK_STEP_2PREV: c_int = 0;
pub const ZK_STEP_3PREV: c_int = 1;
pub const ZK_STEP_3: c_int = 2;
pub const ZK_STEP_42NS: c_int = 3;
pub const ZK_STEP_52NS: c_int = 4;

extern "C" {
    pub fn zk_init(device: c_int) -> c_int;
    pub fn zk_last_error() -> *const c_char;
    pub fn zk_device_count() -> c_int;

    pub fn zk_gl_ntt(src: *const u64, dst: *mut u64, n_pols: u32, nbits: u32, inverse: c_int) -> c_int;
    pub fn zk_gl_lde(src: *const u64, n_pols: u32, nbits: u32, dst: *mut u64, nbits_ext: u32) -> c_int;

    pub fn zk_gl_merkelize(buff: *const u64, width: u32, height: u64) -> *mut zk_merkle_t;
    pub fn zk_merkle_root(t: *const zk_merkle_t, out: *mut u64) -> c_int;
    pub fn zk_merkle_group_proof(t: *const zk_merkle_t, idx: u64, row_out: *mut u64, path_out: *mut u64) -> c_int;
    /// every opening of a query list in one round trip: rows_out[n][width], paths_out[n][depth][4]
    pub fn zk_merkle_group_proofs(t: *const zk_merkle_t, idx: *const u64, n: u32, rows_out: *mut u64, paths_out: *mut u64) -> c_int;
    pub fn zk_merkle_depth(t: *const zk_merkle_t) -> u32;
    pub fn zk_merkle_elements(t: *const zk_merkle_t, out: *mut u64) -> c_int;
    pub fn zk_merkle_free(t: *mut zk_merkle_t) -> c_int;

    pub fn zk_transcript_new() -> *mut zk_transcript_t;
    pub fn zk_transcript_put(t: *mut zk_transcript_t, words: *const u64, n: usize) -> c_int;
    pub fn zk_transcript_get_field(t: *mut zk_transcript_t, out3: *mut u64) -> c_int;
    pub fn zk_transcript_get_fields1(t: *mut zk_transcript_t, out: *mut u64) -> c_int;
    pub fn zk_transcript_get_permutations(t: *mut zk_transcript_t, n: u32, nbits: u32, out: *mut u64) -> c_int;
    pub fn zk_transcript_free(t: *mut zk_transcript_t) -> c_int;

    pub fn zk_stark_setup_new(
        starkinfo_program_json: *const c_char,
        stark_struct_json: *const c_char,
        const_pols: *const u64,
        n_words: u64,
    ) -> *mut zk_stark_setup_t;
    pub fn zk_stark_setup_const_root(s: *const zk_stark_setup_t, out: *mut u64) -> c_int;
    pub fn zk_stark_setup_set_prover_addr(s: *mut zk_stark_setup_t, prover_addr: *const c_char) -> c_int;
    /// stark_verify.rs:20-136 on the zkin text: 1 accepted, 0 rejected, -1 malformed input
    pub fn zk_stark_verify(s: *const zk_stark_setup_t, zkin_json: *const c_char) -> c_int;
    pub fn zk_stark_verify_with(starkinfo_program_json: *const c_char, stark_struct_json: *const c_char, const_root: *const u64, zkin_json: *const c_char) -> c_int;
    /// 1: check 16-ary Merkle paths as loosely as merklehash_bn128.rs:108-128 (default 0 = strict: every level linked); returns the old setting
    pub fn zk_stark_verify_set_reference_compat(on: c_int) -> c_int;
    /// every later zk_stark_gen* verifies its own proof first (prove.rs:124-132)
    pub fn zk_stark_setup_set_self_check(s: *mut zk_stark_setup_t, on: c_int) -> c_int;
    pub fn zk_stark_gen(s: *mut zk_stark_setup_t, cm_pols: *const u64, n_words: u64) -> *mut c_char;
    /// the trace already in HBM, on a stream of the caller's: setups on different streams prove side by side from
    /// different host threads (one proof at a time per setup)
    pub fn zk_stark_gen_dev_on(s: *mut zk_stark_setup_t, d_cm_pols: *const u64, n_words: u64, stream: *mut c_void) -> *mut c_char;
    // ---- the staged prover: what a stark_gen.rs that stays in Rust calls where it calls calculate_exps_parallel (stark_gen.rs:786-792),
    // extend_and_merkelize (:709-750), transcript.get_field, calculate_H1H2 / calculate_Z, the evaluations and FRI::prove (fri.rs:84-89)
    pub fn zk_stark_new(s: *mut zk_stark_setup_t, cm_pols: *const u64, d_cm_pols: *const u64, n_words: u64, stream: *mut c_void) -> *mut zk_stark_ctx_t;
    pub fn zk_stark_commit_stage(c: *mut zk_stark_ctx_t, stage: c_int, root: *mut u64) -> c_int;
    pub fn zk_stark_challenge(c: *mut zk_stark_ctx_t, i: c_int, out: *mut u64) -> c_int;
    pub fn zk_stark_set_challenge(c: *mut zk_stark_ctx_t, i: c_int, v: *const u64) -> c_int;
    pub fn zk_stark_eval(c: *mut zk_stark_ctx_t, step: c_int) -> c_int;
    pub fn zk_stark_calculate_h1h2(c: *mut zk_stark_ctx_t) -> c_int;
    pub fn zk_stark_calculate_z(c: *mut zk_stark_ctx_t) -> c_int;
    pub fn zk_stark_evals(c: *mut zk_stark_ctx_t, evals_out: *mut u64, cap_words: u64) -> c_int;
    pub fn zk_stark_fri_prove(c: *mut zk_stark_ctx_t) -> c_int;
    pub fn zk_stark_finish(c: *mut zk_stark_ctx_t) -> *mut c_char;
    pub fn zk_stark_fri_pol_dev(c: *const zk_stark_ctx_t) -> *const u64;
    pub fn zk_stark_tree(c: *const zk_stark_ctx_t, j: c_int) -> *const zk_merkle_t;
    pub fn zk_stark_free(c: *mut zk_stark_ctx_t) -> c_int;
    /// FRI::prove(transcript, pol, query_pol) alone: the caller's TranscriptGL, a device polynomial, the trees its queries open
    pub fn zk_fri_prove_dev(transcript: *mut zk_transcript_t, d_pol: *const u64, nbits_ext: u32, steps: *const u32, n_steps: u32, n_queries: u32,
                            query_trees: *const *const zk_merkle_t, n_query_trees: u32, stream: *mut c_void) -> *mut c_char;
    pub fn zk_stream_new() -> *mut c_void;
    pub fn zk_stream_sync(stream: *mut c_void) -> c_int;
    pub fn zk_stream_free(stream: *mut c_void) -> c_int;
    pub fn zk_string_free(s: *mut c_char);
    pub fn zk_stark_setup_free(s: *mut zk_stark_setup_t) -> c_int;
    pub fn zk_bn128_load_constants(path: *const c_char) -> c_int;
    pub fn zk_bls12381_load_constants(path: *const c_char) -> c_int;
}

/// `int` status + `zk_last_error()` -> `anyhow::Result` (the reference's error type on these paths)
pub fn check(rc: c_int) -> anyhow::Result<()> {
    if rc == 0 {
        Ok(())
    } else {
        Err(last_error())
    }
}

pub fn last_error() -> anyhow::Error {
    let msg = unsafe { CStr::from_ptr(zk_last_error()) }.to_string_lossy().into_owned();
    anyhow::anyhow!("libzkgpu: {msg}")
}

/// A non-blocking HIP stream of the library's making: one per prover thread (`zk_stark_gen_dev_on`).  Not `Send`: the
/// library orders buffer reuse against the streams of the thread that frees, so a stream stays with the thread that made it.
pub struct Stream(pub *mut c_void);
impl Stream {
    pub fn new() -> anyhow::Result<Self> {
        let p = unsafe { zk_stream_new() };
        if p.is_null() {
            Err(last_error())
        } else {
            Ok(Stream(p))
        }
    }
    pub fn sync(&self) -> anyhow::Result<()> {
        check(unsafe { zk_stream_sync(self.0) })
    }
}
impl Drop for Stream {
    fn drop(&mut self) {
        unsafe { zk_stream_free(self.0) };
    }
}

#[allow(dead_code)]
pub type Opaque = c_void;
