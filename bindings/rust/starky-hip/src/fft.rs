//! Seam 1: `starky::fft_p::{fft, ifft, interpolate}` (starky/src/fft_p.rs:242-261).  Same signatures, same contract: row-major
//! `[1 << nbits][n_pols]`, natural order in and out, `buffdst` sized by the caller, LDE on the coset 49·<w_ext>.
use crate::hip_ffi as ffi;
use fields::field_gl::Fr as FGL;
use starky::traits::FieldExtension;

fn to_words<F: FieldExtension>(v: &[F]) -> Vec<u64> {
    v.iter().map(|e| e.as_int()).collect()
}

fn from_words<F: FieldExtension>(w: &[u64], dst: &mut [F]) {
    for (d, x) in dst.iter_mut().zip(w) {
        *d = F::from(FGL::from(*x));
    }
}

pub fn fft<F: FieldExtension>(buffsrc: &Vec<F>, n_pols: usize, nbits: usize, buffdst: &mut Vec<F>) {
    transform(buffsrc, n_pols, nbits, buffdst, false)
}

pub fn ifft<F: FieldExtension>(buffsrc: &Vec<F>, n_pols: usize, nbits: usize, buffdst: &mut Vec<F>) {
    transform(buffsrc, n_pols, nbits, buffdst, true)
}

fn transform<F: FieldExtension>(buffsrc: &Vec<F>, n_pols: usize, nbits: usize, buffdst: &mut Vec<F>, inverse: bool) {
    let src = to_words(buffsrc);
    let mut dst = vec![0u64; src.len()];
    let rc = unsafe { ffi::zk_gl_ntt(src.as_ptr(), dst.as_mut_ptr(), n_pols as u32, nbits as u32, inverse as i32) };
    ffi::check(rc).expect("zk_gl_ntt");                         // the reference's fft_p panics on misuse as well
    from_words(&dst, buffdst);
}

pub fn interpolate<F: FieldExtension>(buffsrc: &Vec<F>, n_pols: usize, nbits: usize, buffdst: &mut Vec<F>, nbitsext: usize) {
    if buffsrc.is_empty() {
        return;                                                  // fft_p.rs:262-264
    }
    let src = to_words(buffsrc);
    let mut dst = vec![0u64; (1usize << nbitsext) * n_pols];
    let rc = unsafe { ffi::zk_gl_lde(src.as_ptr(), n_pols as u32, nbits as u32, dst.as_mut_ptr(), nbitsext as u32) };
    ffi::check(rc).expect("zk_gl_lde");
    from_words(&dst, buffdst);
}
