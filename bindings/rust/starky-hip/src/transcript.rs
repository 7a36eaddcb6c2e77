//! Seam 4: `impl Transcript` (starky/src/traits.rs:57-63) on the device-resident sponge -- the drop-in for `TranscriptGL`
//! (starky/src/transcript.rs:8-103).  Absorption order and the 63-bit query derivation are the library's; this is plumbing.
use crate::hip_ffi as ffi;
use anyhow::Result;
use fields::field_gl::Fr as FGL;
use starky::traits::{FieldExtension, Transcript};

pub struct TranscriptHipGL {
    handle: *mut ffi::zk_transcript_t,
}

unsafe impl Send for TranscriptHipGL {}

impl Drop for TranscriptHipGL {
    fn drop(&mut self) {
        unsafe { ffi::zk_transcript_free(self.handle) };
    }
}

impl Transcript for TranscriptHipGL {
    fn new() -> Self {
        let handle = unsafe { ffi::zk_transcript_new() };
        assert!(!handle.is_null(), "zk_transcript_new: {}", ffi::last_error());
        Self { handle }
    }

    /// transcript.rs:47-52: three squeezes as (c0, c1, c2)
    fn get_field<F: FieldExtension>(&mut self) -> F {
        let mut w = [0u64; 3];
        ffi::check(unsafe { ffi::zk_transcript_get_field(self.handle, w.as_mut_ptr()) }).expect("zk_transcript_get_field");
        F::from_vec(vec![FGL::from(w[0]), FGL::from(w[1]), FGL::from(w[2])])
    }

    fn get_fields1(&mut self) -> Result<FGL> {
        let mut w = 0u64;
        ffi::check(unsafe { ffi::zk_transcript_get_fields1(self.handle, &mut w) })?;
        Ok(FGL::from(w))
    }

    /// transcript.rs:16-33: `es` is flattened, one word per element
    fn put(&mut self, es: &[Vec<FGL>]) -> Result<()> {
        let words: Vec<u64> = es.iter().flatten().map(|e| e.as_int()).collect();
        ffi::check(unsafe { ffi::zk_transcript_put(self.handle, words.as_ptr(), words.len()) })
    }

    fn get_permutations(&mut self, n: usize, nbits: usize) -> Result<Vec<usize>> {
        let mut out = vec![0u64; n];
        ffi::check(unsafe { ffi::zk_transcript_get_permutations(self.handle, n as u32, nbits as u32, out.as_mut_ptr()) })?;
        Ok(out.into_iter().map(|v| v as usize).collect())
    }
}
