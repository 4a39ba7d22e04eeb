//! Safe wrapper over `ffi`: one method per transcript-delimited stage of `Stark::prove` (src/starks.rs:59-169) and
//! `Fri::prove` (src/fri.rs:53-189).  `prove_gpu` in INTEGRATION.md §4 is written against this type; the Fiat–Shamir
//! transcript (nimue) stays on the Rust side of the boundary.
pub mod convert;
pub mod ffi;
// tree.rs, fri_proof.rs and prove.rs are CHILD modules of the reference's merkle.rs / fri.rs / starks.rs (they build types whose
// fields are private there); their headers say where each one goes.  This file and ffi.rs / convert.rs form `crate::gpu`.
use ffi::*;
use std::ffi::CStr;

#[derive(Debug)]
pub enum GpuError {
    Shape(String),        // -1: the conditions the reference assert!s / panics on (merkle.rs:93-104, air.rs:23, starks.rs:317-320)
    LeafNotFound,         // -2: MerkleProofError::LeafNotFound   (src/error.rs:13-21)
    OutOfRange,           // -3: MerkleProofError::OutOfRangeError
    Backend(i32, String), // state / argument / HIP / memory
}
pub type Digest = [u8; 32];

pub struct Gpu {
    ctx: *mut ms_ctx,
    pub ext: usize, // limbs per extension element: 2 (Goldilocks) / 4 (BabyBear)
}
unsafe impl Send for Gpu {} // one context per host thread (blocking calls, not thread-safe: SURVEY.md 8(b))

impl Gpu {
    pub fn new(device: i32, field: ms_field, flags: u32) -> Result<Self, GpuError> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { ms_create(&mut ctx, device, field, flags) };
        if rc != 0 { return Err(GpuError::Backend(rc, "ms_create failed (no usable GPU / HIP runtime)".into())); }
        let ext = unsafe { ms_ext_degree(ctx) } as usize;
        Ok(Gpu { ctx, ext })
    }
    fn check(&self, rc: i32) -> Result<(), GpuError> {
        if rc == 0 { return Ok(()); }
        let msg = unsafe { CStr::from_ptr(ms_last_error(self.ctx)) }.to_string_lossy().into_owned();
        Err(match rc { -1 => GpuError::Shape(msg), -2 => GpuError::LeafNotFound, -3 => GpuError::OutOfRange, _ => GpuError::Backend(rc, msg) })
    }
    /// starks.rs:68-73: MerkleTree::new(trace.get_data(), lpn).root(); uploads the N x w row-major trace (canonical limbs)
    pub fn trace_commit(&mut self, trace: &[u64], n: usize, w: usize, lpn: usize) -> Result<Digest, GpuError> {
        assert_eq!(trace.len(), n * w);
        let mut root = [0u8; 32];
        self.check(unsafe { ms_trace_commit(self.ctx, trace.as_ptr(), n, w, lpn, root.as_mut_ptr()) })?;
        Ok(root)
    }
    /// Optional, for a pipelined prover: queues the upload of the NEXT proof's trace (page-locked memory from `ms_pinned_alloc`) on a copy engine while the
    /// current proof computes; the `trace_commit` that later passes the same slice finds it on the device.  A hint: `Ok(())` also when nothing was queued.
    /// SAFETY of the contract: `trace` must stay alive and unchanged until that `trace_commit` has returned.
    pub fn trace_upload_async(&mut self, trace: &[u64], n: usize, w: usize) -> Result<(), GpuError> {
        assert_eq!(trace.len(), n * w);
        self.check(unsafe { ms_trace_upload_async(self.ctx, trace.as_ptr(), n, w) })
    }
    /// air.rs:147-160: TraceTable::get_trace_polys (per-column INTT)
    pub fn interpolate(&mut self) -> Result<(), GpuError> { self.check(unsafe { ms_interpolate(self.ctx) }) }
    /// air.rs:127-144: a transition closure that is a linear combination of earlier polynomials (tests/e2e_goldilocks.rs:48-59)
    pub fn polys_lincomb(&mut self, scalars: &[u64], idx: &[i32]) -> Result<(), GpuError> {
        assert_eq!(scalars.len(), idx.len());
        self.check(unsafe { ms_polys_lincomb(self.ctx, scalars.as_ptr(), idx.as_ptr(), idx.len() as i32) })
    }
    /// ... or any closure's output, uploaded as coefficients (at most N of them: starks.rs:118-119 asserts)
    pub fn polys_append(&mut self, coeffs: &[u64]) -> Result<(), GpuError> { self.check(unsafe { ms_polys_append(self.ctx, coeffs.as_ptr(), coeffs.len()) }) }
    pub fn poly_read(&mut self, i: i32, n: usize) -> Result<Vec<u64>, GpuError> {
        let mut out = vec![0u64; n];
        self.check(unsafe { ms_poly_read(self.ctx, i, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// starks.rs:80-95: coset LDE of every constraint polynomial + MerkleTree::new over the L x c matrix
    pub fn lde_commit(&mut self, blowup: usize, shift: u64, lpn: usize) -> Result<Digest, GpuError> {
        let mut root = [0u8; 32];
        self.check(unsafe { ms_lde_commit(self.ctx, blowup, shift, lpn, root.as_mut_ptr()) })?;
        Ok(root)
    }
    /// starks.rs:108-119: validity = sum_i r^i f_i
    pub fn mix(&mut self, r: u64) -> Result<(), GpuError> { self.check(unsafe { ms_mix(self.ctx, r) }) }
    /// starks.rs:124-151: out[t] = (f_0(z_t), .., f_{c-1}(z_t), validity(z_t)), E limbs each
    pub fn eval_ext(&mut self, z_limbs: &[u64], c: usize) -> Result<Vec<u64>, GpuError> {
        let q = z_limbs.len() / self.ext;
        let mut out = vec![0u64; q * (c + 1) * self.ext];
        self.check(unsafe { ms_eval_ext(self.ctx, z_limbs.as_ptr(), q as i32, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// fri.rs:73-82
    pub fn fri_begin(&mut self, blowup: usize, rounds: usize) -> Result<Digest, GpuError> {
        let mut root = [0u8; 32];
        self.check(unsafe { ms_fri_begin(self.ctx, blowup, rounds, root.as_mut_ptr()) })?;
        Ok(root)
    }
    /// fri.rs:89-94: z -> B = [even(z), odd(z)]
    pub fn fri_deep(&mut self, z_limbs: &[u64]) -> Result<Vec<u64>, GpuError> {
        let mut b = vec![0u64; 2 * self.ext];
        self.check(unsafe { ms_fri_deep(self.ctx, z_limbs.as_ptr(), b.as_mut_ptr()) })?;
        Ok(b)
    }
    /// fri.rs:96-109: alpha -> fold, DEEP quotient, next FriRound, its root
    pub fn fri_fold_commit(&mut self, alpha_limbs: &[u64]) -> Result<Digest, GpuError> {
        let mut root = [0u8; 32];
        self.check(unsafe { ms_fri_fold_commit(self.ctx, alpha_limbs.as_ptr(), root.as_mut_ptr()) })?;
        Ok(root)
    }
    /// fri.rs:115-189: the serialised FriProof (MSFP layout: include/ministark.h)
    pub fn fri_query(&mut self, betas: &[u64]) -> Result<Vec<u8>, GpuError> {
        self.check(unsafe { ms_fri_query(self.ctx, betas.as_ptr(), betas.len() as i32) })?;
        let mut blob = vec![0u8; unsafe { ms_fri_proof_size(self.ctx) }];
        self.check(unsafe { ms_fri_proof_read(self.ctx, blob.as_mut_ptr()) })?;
        Ok(blob)
    }
    /// merkle.rs:81-148 on its own (the `impl Tree` of INTEGRATION.md §5 calls this): all nodes, level-major, root last
    pub fn merkle_commit(&mut self, leaf_limbs: &[u64], leaf_num: usize, ext: i32, lpn: usize, ic: usize) -> Result<(Vec<u8>, Digest), GpuError> {
        let mut nn = 0usize;
        let mut root = [0u8; 32];
        let cap = 2 * (leaf_num / lpn) * 32;
        let mut nodes = vec![0u8; cap];
        self.check(unsafe { ms_merkle_commit(self.ctx, leaf_limbs.as_ptr(), leaf_num, ext, lpn, ic, nodes.as_mut_ptr(), cap, &mut nn, root.as_mut_ptr()) })?;
        nodes.truncate(nn * 32);
        Ok((nodes, root))
    }
}
impl Drop for Gpu {
    fn drop(&mut self) { unsafe { ms_destroy(self.ctx) } }
}
