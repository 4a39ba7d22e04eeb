//! `extern "C"` mirror of include/ministark.h, one declaration per exported symbol, in the header's order.
//! Status codes: 0 OK, -1 SHAPE (the reference's assert!/panic! sites), -2 LEAF_NOT_FOUND, -3 OUT_OF_RANGE (src/error.rs:13-21),
//! -4 STATE, -5 ARG, -6 HIP, -7 NOMEM.  Nothing unwinds across this boundary.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_long, c_void};

#[repr(C)]
pub struct ms_ctx {
    _private: [u8; 0],
}
pub type ms_field = c_int;
pub const MS_FIELD_GOLDILOCKS: ms_field = 0; // field.rs:36-56
pub const MS_FIELD_BABYBEAR: ms_field = 1; // field.rs:66-109
pub const MS_FLAG_ZERO_DISPLAY_EMPTY: u32 = 1; // ark-ff 0.5 prints Fp::ZERO as ""
pub const MS_FLAG_TRACE_MONT64: u32 = 2; // ms_trace_commit* reads arkworks memory (Montgomery form, R = 2^64)
pub const MS_FLAG_LATENCY: u32 = 4; // one proof alone on the GPU: independent chains of a stage on two streams, the calling thread spins for a stage's results
pub const MS_FLAGS_DEFAULT: u32 = MS_FLAG_ZERO_DISPLAY_EMPTY;
pub type ms_exchange_fn = Option<unsafe extern "C" fn(user: *mut c_void, op: c_int, bytes: usize) -> c_int>;

extern "C" {
    // ---- context
    pub fn ms_create(out: *mut *mut ms_ctx, device: c_int, field: ms_field, flags: u32) -> c_int;
    pub fn ms_destroy(ctx: *mut ms_ctx);
    pub fn ms_last_error(ctx: *const ms_ctx) -> *const c_char;
    pub fn ms_ext_degree(ctx: *const ms_ctx) -> c_int;
    pub fn ms_set_stream(ctx: *mut ms_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn ms_synchronize(ctx: *mut ms_ctx) -> c_int;
    pub fn ms_pinned_alloc(bytes: usize) -> *mut c_void;
    pub fn ms_pinned_free(p: *mut c_void);
    // ---- one proof over the GPUs of a node
    pub fn ms_set_shard(ctx: *mut ms_ctx, rank: c_int, world: c_int, d_send: *mut c_void, d_recv: *mut c_void, cap_bytes: usize,
                        f: ms_exchange_fn, user: *mut c_void) -> c_int;
    pub fn ms_shard_slice_layout(ctx: *mut ms_ctx, offset: *mut usize, stride: *mut usize) -> c_int;
    pub fn ms_rccl_unique_id(out: *mut u8 /* [128] */) -> c_int;
    pub fn ms_set_shard_rccl(ctx: *mut ms_ctx, rank: c_int, world: c_int, unique_id: *const u8 /* [128] */, cap_bytes: usize) -> c_int;
    pub fn ms_rccl_selftest(ctx: *mut ms_ctx) -> c_int;
    pub fn ms_shard_stats(ctx: *mut ms_ctx, out: *mut u64 /* [8] */) -> c_int;
    pub fn ms_shard_proof_on_root(ctx: *mut ms_ctx, on: c_int) -> c_int;
    pub fn ms_shard_proof_is_elsewhere(ctx: *const ms_ctx) -> c_int;
    pub fn ms_shard_round_is_distributed(ctx: *mut ms_ctx, round: c_int) -> c_int;
    // ---- src/util.rs:4-44, src/starks.rs:268-332
    pub fn ms_is_power_of_two(n: u64) -> c_int;
    pub fn ms_logarithm_of_two_k(n: u64, base: u64) -> c_long;
    pub fn ms_ceil_log2_k(n: u64, base: u64) -> u64;
    pub fn ms_num_queries(f: ms_field, security_bits: u64, blowup: u64, steps: u64, linking_queries: *mut u64, fri_queries_per_round: *mut u64) -> c_int;
    pub fn ms_root_of_unity(f: ms_field, n: u64) -> u64;
    // ---- Stark::prove stages (src/starks.rs:59-169)
    pub fn ms_trace_commit(ctx: *mut ms_ctx, trace_rowmajor: *const u64, n: usize, w: usize, lpn: usize, root: *mut u8 /* [32] */) -> c_int;
    pub fn ms_trace_commit_device(ctx: *mut ms_ctx, d_trace_rowmajor: *const c_void, n: usize, w: usize, lpn: usize, root: *mut u8) -> c_int;
    pub fn ms_trace_upload_async(ctx: *mut ms_ctx, trace_rowmajor: *const u64, n: usize, w: usize) -> c_int;
    pub fn ms_interpolate(ctx: *mut ms_ctx) -> c_int;
    pub fn ms_polys_lincomb(ctx: *mut ms_ctx, scalars: *const u64, idx: *const c_int, k: c_int) -> c_int;
    pub fn ms_polys_append(ctx: *mut ms_ctx, coeffs: *const u64, n: usize) -> c_int;
    pub fn ms_polys_count(ctx: *const ms_ctx) -> c_int;
    pub fn ms_poly_read(ctx: *mut ms_ctx, i: c_int, out: *mut u64) -> c_int;
    pub fn ms_lde_commit(ctx: *mut ms_ctx, blowup: usize, shift: u64, lpn: usize, root: *mut u8) -> c_int;
    pub fn ms_lde_read(ctx: *mut ms_ctx, out_rowmajor: *mut u64) -> c_int;
    pub fn ms_mix(ctx: *mut ms_ctx, r: u64) -> c_int;
    pub fn ms_validity_read(ctx: *mut ms_ctx, out: *mut u64) -> c_int;
    pub fn ms_mix_cubic(ctx: *mut ms_ctx, r: u64, spec: *const c_int, s: *const u64, ncons: c_int) -> c_int;
    pub fn ms_validity_len(ctx: *const ms_ctx) -> usize;
    pub fn ms_eval_ext(ctx: *mut ms_ctx, z: *const u64, q: c_int, out: *mut u64) -> c_int;
    // ---- Fri::prove stages (src/fri.rs:53-189)
    pub fn ms_fri_begin(ctx: *mut ms_ctx, blowup: usize, rounds: usize, root0: *mut u8) -> c_int;
    pub fn ms_fri_deep(ctx: *mut ms_ctx, z: *const u64, b: *mut u64) -> c_int;
    pub fn ms_fri_fold_commit(ctx: *mut ms_ctx, alpha: *const u64, root: *mut u8) -> c_int;
    pub fn ms_fri_round_info(ctx: *mut ms_ctx, round: c_int, ncoef: *mut u64, domain_size: *mut u64) -> c_int;
    pub fn ms_fri_round_poly_read(ctx: *mut ms_ctx, round: c_int, out: *mut u64) -> c_int;
    pub fn ms_fri_round_codeword_read(ctx: *mut ms_ctx, round: c_int, out: *mut u64) -> c_int;
    pub fn ms_fri_query(ctx: *mut ms_ctx, betas: *const u64, nq: c_int) -> c_int;
    pub fn ms_fri_query_into(ctx: *mut ms_ctx, betas: *const u64, nq: c_int, out: *mut u8, cap: usize, len: *mut usize) -> c_int;
    pub fn ms_fri_proof_size(ctx: *const ms_ctx) -> usize;
    pub fn ms_fri_proof_read(ctx: *mut ms_ctx, out: *mut u8) -> c_int;
    pub fn ms_fri_proof_read_async(ctx: *mut ms_ctx, out: *mut u8) -> c_int;
    pub fn ms_fri_proof_wait(ctx: *mut ms_ctx) -> c_int;
    pub fn ms_io_engine(ctx: *const ms_ctx) -> c_int;
    pub fn ms_io_runtime_path() -> *const c_char;
    // ---- Tree trait (src/merkle.rs:8-30) on its own
    pub fn ms_merkle_commit(ctx: *mut ms_ctx, leafs: *const u64, leaf_num: usize, ext: c_int, lpn: usize, ic: usize,
                            nodes_out: *mut u8, nodes_cap: usize, nnodes: *mut usize, root: *mut u8) -> c_int;
    pub fn ms_merkle_prove(ctx: *mut ms_ctx, leafs: *const u64, leaf_num: usize, ext: c_int, lpn: usize, leaf: *const u64,
                           path_out: *mut u8, cap: usize, path_len: *mut usize) -> c_int;
    // ---- standalone transforms and measurement aids
    pub fn ms_ntt(ctx: *mut ms_ctx, data: *mut u64, n: usize, batch: usize, inverse: c_int) -> c_int;
    pub fn ms_coset_lde(ctx: *mut ms_ctx, coeffs: *const u64, ncoef: usize, batch: usize, shift: u64, out: *mut u64, l: usize) -> c_int;
    pub fn ms_bench_lde(ctx: *mut ms_ctx, blowup: usize, shift: u64) -> c_int;
    pub fn ms_arith_selftest(ctx: *mut ms_ctx, op: c_int, a: *const u64, b: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn ms_profile_begin(ctx: *mut ms_ctx) -> c_int;
    pub fn ms_profile_end(ctx: *mut ms_ctx, json_out: *mut c_char, cap: usize) -> c_int;
}
