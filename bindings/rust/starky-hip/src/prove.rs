//! Seam 5: `starky::prove::stark_prove` (starky/src/prove.rs:30-160) with the proof generated on the GPU.
//!
//! What stays on the host, in the reference's own code: reading the PIL / constant / committed files, the code generator
//! (`StarkInfo::new`, starky/src/starkinfo.rs:160-272 -- through `StarkSetup::new`, whose constant tree is the only wasted
//! work and disappears once `StarkSetup` grows a constructor that skips it), `pil2circom`, and `stark_verify` on the result
//! (the reference asserts it, prove.rs:132-140).  What moves: `StarkProof::stark_gen` + `FRI::prove` -> `zk_stark_gen`, whose
//! input is the serde form of `StarkInfo` and `Program` (both derive `Serialize`, starkinfo.rs:27,46) and whose output is the
//! zkin JSON the reference's serializer writes (serializer.rs:146-261), so the file handed to circom is unchanged.
use crate::hip_ffi as ffi;
use anyhow::{anyhow, Result};
use fields::field_gl::Fr as FGL;
use starky::merklehash::MerkleTreeGL;
use starky::merklehash_bls12381::MerkleTreeBLS12381;
use starky::merklehash_bn128::MerkleTreeBN128;
use starky::pil2circom;
use starky::polsarray::{PolKind, PolsArray};
use starky::stark_setup::StarkSetup;
use starky::traits::MerkleTree;
use starky::types::{load_json, StarkStruct, PIL};
use std::ffi::{CStr, CString};
use std::fs::File;
use std::io::Write;

/// Same arguments as `starky::prove::stark_prove`; zkit's `stark_prove` sub-command calls it when built with `--features hip`.
#[allow(clippy::too_many_arguments)]
pub fn stark_prove(
    stark_struct: &str,
    pil_file: &str,
    norm_stage: bool,
    skip_main: bool,
    agg_stage: bool,
    const_pol_file: &str,
    cm_pol_file: &str,
    circom_file: &str,
    zkin: &str,
    prover_addr: &str,
) -> Result<()> {
    let mut pil = load_json::<PIL>(pil_file)?;
    let mut const_pol = PolsArray::new(&pil, PolKind::Constant);
    const_pol.load(const_pol_file)?;
    let mut cm_pol = PolsArray::new(&pil, PolKind::Commit);
    cm_pol.load(cm_pol_file)?;
    let ss = load_json::<StarkStruct>(stark_struct)?;
    let circom_w = File::create(circom_file)?;
    let zkin_w = File::create(zkin)?;
    match ss.verificationHashType.as_str() {
        "GL" => prove::<FGL, MerkleTreeGL, _, _>(&mut pil, const_pol, cm_pol, &ss, agg_stage, norm_stage, skip_main, circom_w, zkin_w, prover_addr),
        "BN128" => {
            load_tables("bn128")?;
            prove::<starky::field_bn128::Fr, MerkleTreeBN128, _, _>(&mut pil, const_pol, cm_pol, &ss, false, norm_stage, skip_main, circom_w, zkin_w, prover_addr)
        }
        "BLS12381" => {
            load_tables("bls12381")?;
            prove::<starky::field_bls12381::Fr, MerkleTreeBLS12381, _, _>(&mut pil, const_pol, cm_pol, &ss, false, norm_stage, skip_main, circom_w, zkin_w, prover_addr)
        }
        other => panic!("Invalid hashtype {other}"),             // prove.rs:89
    }
}

/// the scalar-field Poseidon tables (eigen-zkvm_amd/data/poseidon_<field>_constants.bin), located next to the library
fn load_tables(field: &str) -> Result<()> {
    let dir = std::env::var("ZKGPU_DATA_DIR").map_err(|_| anyhow!("set ZKGPU_DATA_DIR to eigen-zkvm_amd/data"))?;
    let path = CString::new(format!("{dir}/poseidon_{field}_constants.bin"))?;
    let rc = unsafe {
        if field == "bn128" {
            ffi::zk_bn128_load_constants(path.as_ptr())
        } else {
            ffi::zk_bls12381_load_constants(path.as_ptr())
        }
    };
    ffi::check(rc)
}

#[allow(clippy::too_many_arguments)]
fn prove<F, M, W1, W2>(
    pil: &mut PIL,
    const_pol: PolsArray,
    cm_pol: PolsArray,
    stark_struct: &StarkStruct,
    agg_stage: bool,
    norm_stage: bool,
    skip_main: bool,
    mut circom_file_writer: W1,
    mut zkin_writer: W2,
    prover_addr: &str,
) -> Result<()>
where
    F: ff::PrimeField + Default,
    M: MerkleTree<MTNode = starky::ElementDigest<4, F>> + Default,
    W1: Write,
    W2: Write,
{
    // host: code generation (and, for now, the reference's own constant tree) -- prove.rs:108
    let mut setup = StarkSetup::<M>::new(&const_pol, pil, stark_struct, None)?;

    // device: StarkSetup::new's LDE + Merkle of the constants, then stark_gen + FRI::prove
    let program = CString::new(format!(
        "{{\"starkinfo\":{},\"program\":{}}}",
        serde_json::to_string(&setup.starkinfo)?,
        serde_json::to_string(&setup.program)?
    ))?;
    let ss_json = CString::new(serde_json::to_string(stark_struct)?)?;
    let words = |p: &PolsArray| -> Vec<u64> { p.array.iter().flatten().map(|e| e.as_int()).collect() }; // row-major, polsarray.rs:219-227
    let (const_w, cm_w) = (words(&const_pol), words(&cm_pol));
    ffi::check(unsafe { ffi::zk_init(0) })?;
    let h = unsafe { ffi::zk_stark_setup_new(program.as_ptr(), ss_json.as_ptr(), const_w.as_ptr(), const_w.len() as u64) };
    if h.is_null() {
        return Err(ffi::last_error());
    }
    let addr = CString::new(prover_addr)?;
    ffi::check(unsafe { ffi::zk_stark_setup_set_prover_addr(h, addr.as_ptr()) })?;
    let p = unsafe { ffi::zk_stark_gen(h, cm_w.as_ptr(), cm_w.len() as u64) };
    if p.is_null() {
        unsafe { ffi::zk_stark_setup_free(h) };
        return Err(ffi::last_error());
    }
    let zkin_json = unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned();
    unsafe {
        ffi::zk_string_free(p);
        ffi::zk_stark_setup_free(h);
    }

    // host: the reference checks its own proof before writing anything (prove.rs:132-140); the zkin JSON deserialises into
    // StarkProof<M> through the reference's serializer (serializer.rs:264-420)
    let proof: starky::stark_gen::StarkProof<M> = serde_json::from_str(&zkin_json)?;
    let ok = starky::stark_verify::stark_verify::<M, <M as DefaultTranscript>::T>(
        &proof,
        &setup.const_root,
        &setup.starkinfo,
        stark_struct,
        &setup.program,
    )?;
    assert!(ok);

    let opt = pil2circom::StarkOption { enable_input: false, verkey_input: norm_stage, skip_main, agg_stage };
    let str_ver = pil2circom::pil2circom::<F>(pil, &setup.const_root, stark_struct, &mut setup.starkinfo, &mut setup.program, &opt)?;
    write!(circom_file_writer, "{str_ver}")?;
    write!(zkin_writer, "{zkin_json}")?;
    Ok(())
}

/// the transcript each tree type is proved with (prove.rs:47-91)
pub trait DefaultTranscript {
    type T: starky::traits::Transcript;
}
impl DefaultTranscript for MerkleTreeGL {
    type T = starky::transcript::TranscriptGL;
}
impl DefaultTranscript for MerkleTreeBN128 {
    type T = starky::transcript_bn128::TranscriptBN128;
}
impl DefaultTranscript for MerkleTreeBLS12381 {
    type T = starky::transcript_bls12381::TranscriptBLS128;
}
