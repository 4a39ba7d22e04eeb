//! `impl Tree for GpuMerkleTree` (the `Tree` trait, src/merkle.rs:8-30, is the one seam of the reference that is
//! already abstract).  Drop-in for `MerkleTree<Sha256, F>` wherever only the commitment and the node array are needed
//! (src/starks.rs:70,92; src/fri.rs:351): `new` runs leaf hashing + all inner levels on the GPU through `ms_merkle_commit`
//! (src/merkle.rs:81-148 -> csrc/merkle.hpp), the nodes come back level-major, root last - exactly `MerkleTree.nodes`.
//!
//! WHERE IT GOES: `mod gpu_tree;` inside the reference's src/merkle.rs (a child module, so that `MerklePath`'s private fields are
//! visible to `merkle_path_from_parts` below); `Gpu` / `ffi` are this crate's lib.rs / ffi.rs.
use super::{Hash, MerklePath, MerkleTreeConfig, Tree};
use crate::gpu::convert::to_limbs;
use crate::gpu::{ffi, Gpu};
use ark_ff::FftField;
use digest::Digest;
use sha2::Sha256;
use std::cell::RefCell;

thread_local! {
    /// one context per host thread (the ABI is blocking and not thread-safe: SURVEY.md 8(b)); created on first use for the field of `F`
    static GPU: RefCell<Option<Gpu>> = RefCell::new(None);
}
fn with_gpu<F: FftField, R>(f: impl FnOnce(&mut Gpu) -> R) -> R {
    GPU.with(|g| {
        let mut g = g.borrow_mut();
        if g.is_none() {
            // 64-bit modulus = Goldilocks (src/field.rs:43-47), 31-bit = BabyBear (src/field.rs:72-76)
            let field = if <F::BasePrimeField as ark_ff::PrimeField>::MODULUS_BIT_SIZE > 32 { ffi::MS_FIELD_GOLDILOCKS } else { ffi::MS_FIELD_BABYBEAR };
            *g = Some(Gpu::new(0, field, ffi::MS_FLAGS_DEFAULT).expect("no usable GPU: GpuMerkleTree has no CPU fallback"));
        }
        f(g.as_mut().unwrap())
    })
}

#[derive(Clone)]
pub struct GpuMerkleTree<D: Digest, F: FftField> {
    leafs: Vec<F>,
    nodes: Vec<Hash<D>>,
    config: MerkleTreeConfig<D, F>,
}

impl<F: FftField> Tree for GpuMerkleTree<Sha256, F> {
    type Input = F;
    type Inner = Hash<Sha256>;
    type Config = MerkleTreeConfig<Sha256, F>;

    /// src/merkle.rs:81-148.  Panics where the reference panics (MS_ERR_SHAPE: merkle.rs:93-104).
    fn new(inputs: &[F], config: Self::Config) -> Self {
        let ext = F::extension_degree() as i32;
        let limbs = to_limbs(inputs);
        let (bytes, _root) = with_gpu::<F, _>(|gpu| gpu.merkle_commit(&limbs, inputs.len(), ext, config.leafs_per_node, config.inner_children))
            .unwrap_or_else(|e| panic!("{:?}", e));
        let nodes = bytes.chunks_exact(32).map(|c| Hash::<Sha256>::clone_from_slice(c)).collect();
        Self { leafs: inputs.to_vec(), nodes, config }
    }
    /// src/merkle.rs:151-154
    fn root(&self) -> Hash<Sha256> { self.nodes.last().unwrap().clone() }
    /// src/merkle.rs:157-159
    fn get_node_number(&self) -> usize { self.leafs.len() + self.nodes.len() }
    /// src/merkle.rs:162-168 - one group on the CPU (the verifier's side of the trait; the GPU path hashes all groups in `new`)
    fn calculate_from_leafs(children: &[F]) -> Hash<Sha256> {
        let mut hasher = Sha256::new();
        for child in children.iter() { hasher.update(child.to_string()); }
        hasher.finalize()
    }
    /// src/merkle.rs:171-177
    fn calculate_from_nodes(children: &[Hash<Sha256>]) -> Hash<Sha256> {
        let mut hasher = Sha256::new();
        for child in children { hasher.update(child) }
        hasher.finalize()
    }
}
impl<D: Digest, F: FftField> GpuMerkleTree<D, F> {
    pub fn nodes(&self) -> &[Hash<D>] { &self.nodes }
    pub fn config(&self) -> &MerkleTreeConfig<D, F> { &self.config }
}

/// `MerklePath` (src/merkle.rs:293-298) from the parts the MSFP blob carries; lives here because `path` is private to `merkle`
pub(crate) fn merkle_path_from_parts<D: Digest, F: FftField>(leaf_neighbours: Vec<F>, path: Vec<Vec<Hash<D>>>) -> MerklePath<D, F> {
    MerklePath { leaf_neighbours, path }
}
