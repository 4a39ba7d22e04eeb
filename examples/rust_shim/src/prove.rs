//! `Stark::prove_gpu`: src/starks.rs:59-169 with every arkworks call of the hot path replaced by the stage function of the C ABI
//! that subsumes it; the Fiat-Shamir transcript (nimue `Merlin`) stays here, message for message in the reference's
//! order, so `StarkProof.arthur` and every challenge are the reference's own.  INTEGRATION.md section 4 walks through it.
//!
//! WHERE IT GOES: `mod gpu_prove;` inside the reference's src/starks.rs (a child module: `Stark`'s config tuple field and
//! `StarkProof`'s fields are private to `starks`).
use super::{Stark, StarkProof};
use crate::air::Provable;
use crate::error::ProverError;
use crate::field::StarkField;
use crate::fri::FriProof;
use crate::gpu::convert::{base_to_u64, from_limbs, push_limbs, to_limbs};
use crate::gpu::{Gpu, GpuError};
use crate::Hash;
use ark_ff::{Field, Zero};
use ark_poly::EvaluationDomain;
use digest::core_api::BlockSizeUser;
use digest::{Digest, FixedOutputReset};
use nimue::plugins::ark::{FieldChallenges, FieldWriter};
use nimue::{ByteChallenges, ByteWriter};

impl From<GpuError> for ProverError {
    fn from(e: GpuError) -> Self {
        match e {
            GpuError::LeafNotFound => ProverError::MerkleProof(crate::error::MerkleProofError::LeafNotFound),       // src/error.rs:13-21
            GpuError::OutOfRange => ProverError::MerkleProof(crate::error::MerkleProofError::OutOfRangeError),
            GpuError::Shape(m) => panic!("{}", m),                                                                 // the reference assert!s / panics here
            GpuError::Backend(rc, m) => panic!("ministark backend error {}: {}", rc, m),
        }
    }
}

/// A transition closure as the ABI takes it: sum_t scalars[t] * P_idx[t] over the trace polynomials (tests/e2e_goldilocks.rs:48-59
/// are of this form); anything else goes through `Gpu::polys_append` as coefficients.
pub struct LinearTransition<B> { pub scalars: Vec<B>, pub idx: Vec<i32> }

impl<D, F> Stark<D, F>
where
    F: StarkField,
    D: Digest + FixedOutputReset + BlockSizeUser + Clone,
{
    pub fn prove_gpu<T, AIR: Provable<T, F::Base>>(
        &self,
        gpu: &mut Gpu,
        air: AIR,
        witness: T,
        transitions: &[LinearTransition<F::Base>],
    ) -> Result<StarkProof<D, F>, ProverError> {
        let e = <F::Extension as Field>::extension_degree() as usize;
        let mut merlin = self.0.io.to_merlin();                                         // starks.rs:64
        // 1.1 trace + commitment (starks.rs:68-73) -> ms_trace_commit
        let trace = air.trace(&witness);
        let n = trace.get_domain().size();
        let w = trace.trace.get_data().len() / n;
        let lpn = self.0.merkle_config.leafs_per_node;
        let rows: Vec<u64> = trace.trace.get_data().iter().map(base_to_u64).collect();  // canonical limbs (MS_REPR canonical)
        let trace_commit = digest_of::<D>(gpu.trace_commit(&rows, n, w, lpn)?);
        merlin.add_bytes(&trace_commit)?;
        // 1.2 coset LDE of the constraint polynomials + commitment (starks.rs:80-95) -> ms_interpolate, ms_polys_lincomb, ms_lde_commit
        let [random_shift]: [F::Base; 1] = merlin.challenge_scalars()?;                 // starks.rs:81
        gpu.interpolate()?;                                                             // air.rs:147-160
        for t in transitions {                                                          // air.rs:130-134
            let sc: Vec<u64> = t.scalars.iter().map(base_to_u64).collect();
            gpu.polys_lincomb(&sc, &t.idx)?;
        }
        let constrain_trace_commit = digest_of::<D>(gpu.lde_commit(self.0.blowup_factor, base_to_u64(&random_shift), lpn)?);
        merlin.add_bytes(&constrain_trace_commit)?;                                     // starks.rs:95
        // 1.3 mix (starks.rs:108-119; quirk Q1: the remainder is the validity polynomial) -> ms_mix
        let [r]: [F::Base; 1] = merlin.challenge_scalars()?;
        gpu.mix(base_to_u64(&r))?;
        // 2. DEEP-ALI (starks.rs:124-151) -> ms_eval_ext
        let mut queries = vec![F::Extension::zero(); self.0.constrain_queries];
        merlin.fill_challenge_scalars(&mut queries)?;
        let c = w + transitions.len();                                                  // air.rs:123-125
        let evals = gpu.eval_ext(&to_limbs(&queries), c)?;                              // [q][c+1][E]
        let (mut constrain_queries, mut validity_queries) = (Vec::new(), Vec::new());
        for t in 0..queries.len() {
            let row = &evals[t * (c + 1) * e..(t + 1) * (c + 1) * e];
            constrain_queries.push((0..c).map(|i| from_limbs::<F::Extension>(&row[i * e..(i + 1) * e])).collect::<Vec<_>>());
            validity_queries.push(from_limbs::<F::Extension>(&row[c * e..(c + 1) * e]));
        }
        // 3. Fri::commit_phase (fri.rs:64-113) -> ms_fri_begin, ms_fri_deep, ms_fri_fold_commit
        let fri = &self.0.fri_config;
        let _root0 = gpu.fri_begin(fri.blowup_factor, fri.rounds)?;                     // fri.rs:73-82 (never sent: as in the reference)
        for _ in 1..fri.rounds {
            let [z]: [F::Extension; 1] = merlin.challenge_scalars()?;                   // fri.rs:89
            let mut zl = Vec::with_capacity(e); push_limbs(&z, &mut zl);
            let b = gpu.fri_deep(&zl)?;                                                 // fri.rs:90-93
            let bs = [from_limbs::<F::Extension>(&b[..e]), from_limbs::<F::Extension>(&b[e..])];
            merlin.add_scalars(&bs)?;                                                   // fri.rs:94
            let [alpha]: [F::Extension; 1] = merlin.challenge_scalars()?;               // fri.rs:96
            let mut al = Vec::with_capacity(e); push_limbs(&alpha, &mut al);
            let root = digest_of::<D>(gpu.fri_fold_commit(&al)?);                       // fri.rs:97-107
            merlin.add_bytes(&root)?;                                                   // fri.rs:108
        }
        // Fri::query_phase (fri.rs:115-189) -> ms_fri_query + ms_fri_proof_read
        let mut raw = vec![0u8; 8 * fri.queries];
        merlin.fill_challenge_bytes(&mut raw)?;                                         // fri.rs:121-122
        let betas: Vec<u64> = raw.chunks_exact(8).map(|a| u64::from_le_bytes(a.try_into().unwrap())).collect();   // usize::from_le_bytes, fri.rs:123-126
        let blob = gpu.fri_query(&betas)?;
        let fri_proof = FriProof::<D, F::Extension>::from_msfp(&blob, fri.rounds, fri.queries).expect("the library wrote this blob");
        let arthur = merlin.transcript().to_vec();                                      // starks.rs:159
        Ok(StarkProof { arthur, trace_commit, constrain_trace_commit, constrain_queries, validity_queries, fri_proof })
    }
}
fn digest_of<D: Digest>(d: [u8; 32]) -> Hash<D> { Hash::<D>::clone_from_slice(&d) }
