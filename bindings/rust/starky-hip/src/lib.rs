//! starky-hip -- eigen-zkvm's STARK prover seams on an AMD MI355X through libzkgpu.so (include/zkgpu.h).
//!
//! | module | reference seam | what it replaces |
//! |---|---|---|
//! | [`fft`] | `starky::fft_p::{fft, ifft, interpolate}` (fft_p.rs:242-261) | batched NTT / LDE |
//! | [`merkle`] | `trait MerkleTree` (traits.rs:24-55) | `MerkleTreeGL` |
//! | [`transcript`] | `trait Transcript` (traits.rs:57-63) | `TranscriptGL` |
//! | [`prove`] | `starky::prove::stark_prove` (prove.rs:30-160) | `StarkProof::stark_gen` + `FRI::prove`, whole proof on the device |
//!
//! The first three let the reference's own generic prover (`StarkProof::<M>::stark_gen::<T>`) run with device trees and
//! transforms, one call at a time; [`prove`] is the fast path (everything resident in HBM between the stages).
pub mod fft;
pub mod hip_ffi;
pub mod merkle;
pub mod prove;
pub mod transcript;

pub use merkle::MerkleTreeHipGL;
pub use transcript::TranscriptHipGL;
