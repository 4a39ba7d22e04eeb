//! Field elements <-> the u64 limbs the C ABI speaks (`MS_REPR` canonical: SURVEY.md 8(b) "data representation trap").
//! A Rust `&[GoldilocksFp]` is Montgomery-form memory and `Fp` / `QuadExtField` are not `repr(C)`, so elements are marshalled
//! through `into_bigint()` (canonical) rather than reinterpreted.  Limb order of an extension element = arkworks'
//! `to_base_prime_field_elements()` order (c0, c1; for Fp4: c0.c0, c0.c1, c1.c0, c1.c1) = the order of the nested
//! `QuadExtField(c0 + c1 * u)` Display the leaf hashes are built from (src/merkle.rs:162-168).
use ark_ff::{BigInteger, Field, PrimeField};

/// canonical value of a one-limb prime-field element (src/field.rs:47,76: `Fp<MontBackend<_, 1>, 1>`)
pub fn base_to_u64<P: PrimeField>(x: &P) -> u64 {
    let b = x.into_bigint();
    debug_assert!(b.as_ref().iter().skip(1).all(|l| *l == 0), "one-limb fields only");
    b.as_ref()[0]
}
pub fn base_from_u64<P: PrimeField>(v: u64) -> P { P::from(v) }

/// E limbs of one (extension) element, appended to `out`
pub fn push_limbs<F: Field>(x: &F, out: &mut Vec<u64>) {
    for l in x.to_base_prime_field_elements() { out.push(base_to_u64(&l)); }
}
pub fn to_limbs<F: Field>(xs: &[F]) -> Vec<u64> {
    let mut out = Vec::with_capacity(xs.len() * F::extension_degree() as usize);
    for x in xs { push_limbs(x, &mut out); }
    out
}
/// one element from its E limbs
pub fn from_limbs<F: Field>(limbs: &[u64]) -> F {
    debug_assert_eq!(limbs.len(), F::extension_degree() as usize);
    F::from_base_prime_field_elems(limbs.iter().map(|l| base_from_u64::<F::BasePrimeField>(*l))).expect("E limbs per element")
}
pub fn vec_from_limbs<F: Field>(limbs: &[u64]) -> Vec<F> {
    let e = F::extension_degree() as usize;
    limbs.chunks_exact(e).map(from_limbs::<F>).collect()
}
