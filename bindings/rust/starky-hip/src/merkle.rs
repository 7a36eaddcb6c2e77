//! Seam 3: `impl MerkleTree` (starky/src/traits.rs:24-55) backed by a device-resident tree -- the drop-in for `MerkleTreeGL`
//! (starky/src/merklehash.rs:237-457) wherever the prover is generic over `M: MerkleTree`.
use crate::hip_ffi as ffi;
use anyhow::{bail, Result};
use fields::field_gl::Fr as FGL;
use starky::f3g::F3G;
use starky::merklehash::MerkleTreeGL;
use starky::traits::{MTNodeType, MerkleTree};
use starky::ElementDigest;

pub struct MerkleTreeHipGL {
    handle: *mut ffi::zk_merkle_t,
    width: usize,
    height: usize,
    host: MerkleTreeGL, // verification of group proofs is a verifier-side, host-only operation
}

// the handle is only touched through &mut self / &self calls that the library serialises per handle
unsafe impl Send for MerkleTreeHipGL {}

impl Default for MerkleTreeHipGL {
    fn default() -> Self {
        <Self as MerkleTree>::new()
    }
}

impl Drop for MerkleTreeHipGL {
    fn drop(&mut self) {
        if !self.handle.is_null() {
            unsafe { ffi::zk_merkle_free(self.handle) };
        }
    }
}

impl MerkleTree for MerkleTreeHipGL {
    type BaseField = FGL;
    type MTNode = ElementDigest<4, FGL>;
    type ExtendField = F3G;

    fn new() -> Self {
        Self { handle: std::ptr::null_mut(), width: 0, height: 0, host: MerkleTreeGL::new() }
    }

    fn element_size(&self) -> usize {
        self.width * self.height
    }

    /// merklehash.rs:260-265: the prover's `to_extend` copies the committed rows back as F3G
    fn to_extend(&self, p_be: &mut Vec<F3G>) {
        assert_eq!(p_be.len(), self.element_size());
        let mut words = vec![0u64; self.element_size()];
        ffi::check(unsafe { ffi::zk_merkle_elements(self.handle, words.as_mut_ptr()) }).expect("zk_merkle_elements");
        for (o, w) in p_be.iter_mut().zip(&words) {
            *o = F3G::from(FGL::from(*w));
        }
    }

    fn to_basefield(node: &Self::MTNode) -> Vec<FGL> {
        vec![node.as_elements()[0]]
    }

    fn from_basefield(node: &FGL) -> Self::MTNode {
        Self::MTNode::new(&[*node, FGL::ZERO, FGL::ZERO, FGL::ZERO])
    }

    /// merklehash.rs:293-346.  The reference takes ownership of `buff` and keeps it as `elements`; here the tree keeps its own
    /// copy in HBM and the host vector is dropped.
    fn merkelize(&mut self, buff: Vec<FGL>, width: usize, height: usize) -> Result<()> {
        if buff.len() != width * height {
            bail!("buff.len() != width * height");
        }
        let words: Vec<u64> = buff.iter().map(|e| e.as_int()).collect();
        drop(buff);
        if !self.handle.is_null() {
            unsafe { ffi::zk_merkle_free(self.handle) };
        }
        self.handle = unsafe { ffi::zk_gl_merkelize(words.as_ptr(), width as u32, height as u64) };
        if self.handle.is_null() {
            return Err(ffi::last_error());
        }
        self.width = width;
        self.height = height;
        Ok(())
    }

    fn get_element(&self, idx: usize, sub_idx: usize) -> FGL {
        let (row, _) = self.get_group_proof(idx).expect("get_element");
        row[sub_idx]
    }

    /// merklehash.rs:430-438: (row values, one sibling digest per level); `idx >= height` is an error there and here
    fn get_group_proof(&self, idx: usize) -> Result<(Vec<FGL>, Vec<Vec<FGL>>)> {
        let depth = unsafe { ffi::zk_merkle_depth(self.handle) } as usize;
        let mut row = vec![0u64; self.width];
        let mut path = vec![0u64; 4 * depth];
        ffi::check(unsafe { ffi::zk_merkle_group_proof(self.handle, idx as u64, row.as_mut_ptr(), path.as_mut_ptr()) })?;
        let row = row.into_iter().map(FGL::from).collect();
        let mp = path.chunks(4).map(|lvl| lvl.iter().map(|w| FGL::from(*w)).collect()).collect();
        Ok((row, mp))
    }

    fn verify_group_proof(&self, root: &Self::MTNode, mp: &[Vec<FGL>], idx: usize, group_elements: &[FGL]) -> Result<bool> {
        self.host.verify_group_proof(root, mp, idx, group_elements) // merklehash.rs:440-453
    }

    fn root(&self) -> Self::MTNode {
        let mut w = [0u64; 4];
        ffi::check(unsafe { ffi::zk_merkle_root(self.handle, w.as_mut_ptr()) }).expect("zk_merkle_root");
        Self::MTNode::new(&[FGL::from(w[0]), FGL::from(w[1]), FGL::from(w[2]), FGL::from(w[3])])
    }

    fn eq_root(&self, r1: &Self::MTNode, r2: &Self::MTNode) -> bool {
        r1 == r2
    }
}
