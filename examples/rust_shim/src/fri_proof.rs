//! `FriProof::from_msfp`: the reference's `FriProof` (src/fri.rs:17-22) out of the serialised blob the query-phase kernels write
//! ("MSFP", layout in include/ministark.h; the reference derives no serialisation, so the layout is build-defined).  The compiled,
//! tested twin of this walk is `msh_fri_proof_parse` in libministark_host.so (tests/test_host_mirror.py).
//!
//! WHERE IT GOES: `mod msfp;` inside the reference's src/fri.rs (a child module: `FriProof`'s fields are private to `fri`).
//!
//! per window i = 0 .. rounds-2, per query j = 0 .. queries-1:
//!     6*E u64   x1 y1 x2 y2 x3 y3                      -> points[i][j] = [(x1,y1),(x2,y2),(x3,y3)]        (fri.rs:148-154)
//!     u64 qlen, qlen*E u64 quotient coefficients       -> quotients[i][j]                                  (fri.rs:159-167)
//!     MerklePath(y1), MerklePath(y2)                   -> queries[i][j]                                    (fri.rs:170-172)
//! MerklePath = u64 leaf_index | 2*E u64 leaf_neighbours | u64 nlevels | nlevels * 2 * 32 bytes            (merkle.rs:272-298)
use super::FriProof;
use crate::gpu::convert::{from_limbs, vec_from_limbs};
use crate::merkle::gpu_tree::merkle_path_from_parts;
use crate::merkle::MerklePath;
use crate::Hash;
use ark_ff::FftField;
use digest::Digest;

#[derive(Debug)]
pub struct MsfpError(pub &'static str);

struct Reader<'a> { b: &'a [u8], pos: usize }
impl<'a> Reader<'a> {
    fn take(&mut self, n: usize) -> Result<&'a [u8], MsfpError> {
        if n > self.b.len() - self.pos { return Err(MsfpError("MSFP blob truncated")); }
        let s = &self.b[self.pos..self.pos + n];
        self.pos += n;
        Ok(s)
    }
    fn u64(&mut self) -> Result<u64, MsfpError> { Ok(u64::from_le_bytes(self.take(8)?.try_into().unwrap())) }
    fn limbs(&mut self, n: usize) -> Result<Vec<u64>, MsfpError> {
        Ok(self.take(8 * n)?.chunks_exact(8).map(|c| u64::from_le_bytes(c.try_into().unwrap())).collect())
    }
    fn path<D: Digest, F: FftField>(&mut self, e: usize) -> Result<MerklePath<D, F>, MsfpError> {
        let _leaf_index = self.u64()?;                       // informative: the reference's MerklePath does not keep it
        let neighbours = vec_from_limbs::<F>(&self.limbs(2 * e)?);   // leafs_per_node = 2 in the FRI trees (starks.rs:290-295)
        let nlevels = self.u64()? as usize;
        if nlevels > 64 { return Err(MsfpError("MSFP: absurd Merkle path length")); }
        let mut path = Vec::with_capacity(nlevels);
        for _ in 0..nlevels {                                // inner_children = 2: the sibling pair of every level (merkle.rs:241-265)
            let pair = self.take(64)?;
            path.push(vec![Hash::<D>::clone_from_slice(&pair[..32]), Hash::<D>::clone_from_slice(&pair[32..])]);
        }
        Ok(merkle_path_from_parts(neighbours, path))
    }
}

impl<D: Digest, F: FftField> FriProof<D, F> {
    /// `rounds` = FriConfig.rounds, `queries` = FriConfig.queries (src/fri.rs:24-30); the blob must parse exactly
    pub fn from_msfp(blob: &[u8], rounds: usize, queries: usize) -> Result<Self, MsfpError> {
        let e = F::extension_degree() as usize;
        let mut r = Reader { b: blob, pos: 0 };
        let windows = rounds.saturating_sub(1);
        let (mut points, mut quotients, mut paths) = (Vec::with_capacity(windows), Vec::with_capacity(windows), Vec::with_capacity(windows));
        for _ in 0..windows {
            let (mut pw, mut qw, mut mw) = (Vec::with_capacity(queries), Vec::with_capacity(queries), Vec::with_capacity(queries));
            for _ in 0..queries {
                let p = r.limbs(6 * e)?;
                let el = |k: usize| from_limbs::<F>(&p[k * e..(k + 1) * e]);
                pw.push([(el(0), el(1)), (el(2), el(3)), (el(4), el(5))]);
                let qlen = r.u64()? as usize;
                if qlen > (1usize << 40) { return Err(MsfpError("MSFP: absurd quotient length")); }
                qw.push(vec_from_limbs::<F>(&r.limbs(qlen * e)?));
                let p1 = r.path::<D, F>(e)?;
                let p2 = r.path::<D, F>(e)?;
                mw.push([p1, p2]);
            }
            points.push(pw); quotients.push(qw); paths.push(mw);
        }
        if r.pos != blob.len() { return Err(MsfpError("trailing bytes in the MSFP blob")); }
        Ok(FriProof { points, queries: paths, quotients })
    }
}
