/* ministark.h — C ABI of the MI355X-native backend for mini-stark's
 * low-degree-extension + FRI + Merkle proving path.
 *
 * The reference (alv-around/mini-stark, Rust on arkworks) has no FFI; the seams
 * this ABI sits behind are `Stark::prove` (src/starks.rs:59-169), `Fri::prove`
 * (src/fri.rs:53-189), the `Tree` trait (src/merkle.rs:8-30) and the arkworks
 * calls they make.  One function per transcript-delimited stage, so the
 * Fiat–Shamir transcript (nimue) stays with the caller: every challenge is an
 * INPUT and every commitment / opened value an OUTPUT.  INTEGRATION.md shows the
 * Rust `extern "C"` block and the patched `prove` that binds these symbols.
 *
 * Conventions
 *  - Field elements cross the boundary as canonical little-endian u64 limbs
 *    (`Fp::into_bigint()`), also for BabyBear (the reference stores one u64
 *    limb for both fields, src/field.rs:47,76).  Extension elements are E
 *    consecutive limbs: Goldilocks Fp2 = (c0, c1); BabyBear Fp4 =
 *    (c0.c0, c0.c1, c1.c0, c1.c1)  (src/field.rs:50-109).
 *  - Digests are 32 raw SHA-256 bytes (`Hash<Sha256>`, src/lib.rs:13).
 *  - Every function returns MS_OK or a negative ms_status; nothing unwinds.
 *    Conditions on which the reference panics/asserts map to MS_ERR_SHAPE
 *    (src/merkle.rs:93-104, src/air.rs:23-26,53-54, src/starks.rs:317-320);
 *    src/error.rs:13-21 maps to MS_ERR_LEAF_NOT_FOUND / MS_ERR_OUT_OF_RANGE.
 *  - Blocking calls; one ms_ctx per host thread and per GPU (the reference is
 *    single-threaded, `Stark::prove(&self)`); not thread-safe.
 *  - Host buffers are caller-owned and only read/written during the call.
 *    Large intermediates (polynomials, LDE, trees, FRI rounds) stay in HBM
 *    inside the context until the next proof or ms_destroy.
 */
#ifndef MINISTARK_H
#define MINISTARK_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ms_ctx ms_ctx;

typedef enum { MS_FIELD_GOLDILOCKS = 0, MS_FIELD_BABYBEAR = 1 } ms_field; /* src/field.rs:36-76 */

typedef enum {
  MS_OK = 0,
  MS_ERR_SHAPE = -1,          /* reference assert!/panic */
  MS_ERR_LEAF_NOT_FOUND = -2, /* MerkleProofError::LeafNotFound, src/error.rs:15-18 */
  MS_ERR_OUT_OF_RANGE = -3,   /* MerkleProofError::OutOfRangeError, src/error.rs:19-21 */
  MS_ERR_STATE = -4,          /* stage called out of order */
  MS_ERR_ARG = -5,            /* null pointer, non-canonical element, unsupported parameter */
  MS_ERR_HIP = -6,            /* HIP runtime failure; see ms_last_error */
  MS_ERR_NOMEM = -7
} ms_status;

/* flags for ms_create */
#define MS_FLAG_ZERO_DISPLAY_EMPTY 0x1u /* ark-ff Display prints ZERO as "" (default behaviour of ark-ff 0.5.0) */
#define MS_FLAG_TRACE_MONT64 0x2u       /* ms_trace_commit* input is arkworks memory: Montgomery form x*2^64 mod p, one u64 limb
                                          (`Fp<MontBackend<_,1>,1>`, src/field.rs:47,76); everything else stays canonical */
#define MS_FLAG_LATENCY 0x4u            /* this context proves ALONE on its GPU and the caller wants the shortest proof, not the most proofs: independent chains of
                                         * a stage (a FRI round's coefficient side and evaluation side) run on two streams, and the calling thread SPINS on a
                                         * page-locked word the stage's last kernel stores behind its results instead of blocking in the stream synchronisation
                                         * (4-5 us less per transcript round trip, 44 of them per proof; a stage longer than 2 ms falls back to the blocking wait).
                                         * With several contexts in flight it costs throughput (the side streams compete for HIP's few hardware queues: -15 % at
                                         * 8 in flight; every proving thread burns a core while it waits): leave it off there. */
#define MS_FLAGS_DEFAULT MS_FLAG_ZERO_DISPLAY_EMPTY

/* ---- context ------------------------------------------------------------ */
int ms_create(ms_ctx** out, int device, ms_field field, uint32_t flags);
void ms_destroy(ms_ctx* ctx);
const char* ms_last_error(const ms_ctx* ctx);
int ms_ext_degree(const ms_ctx* ctx);             /* 2 (Goldilocks) / 4 (BabyBear): StarkField::Extension */
int ms_set_stream(ms_ctx* ctx, void* hip_stream); /* run on a caller-owned hipStream_t (NULL: back to the ctx's own) */
int ms_synchronize(ms_ctx* ctx);
/* page-locked host memory for the boundary's bulk transfers (the trace going in, the FRI proof coming out): copies to and from
 * it are DMA transfers that overlap with kernels of other contexts.  NULL on failure. */
void* ms_pinned_alloc(size_t bytes);
void ms_pinned_free(void* p);

/* ---- one proof sharded over the GPUs of a node (no reference counterpart) ----
 * One process per GPU, every rank calls the SAME stage functions with the SAME inputs and gets the same
 * outputs.  Coefficient-domain work is replicated; the evaluation-domain work of every large commitment
 * (coset LDE / FRI codeword, leaf hashing, Merkle subtree) is partitioned: rank k evaluates and hashes the
 * leaf groups j = k (mod world) (a union of cosets of the evaluation domain: no data moves), then
 *   ALL_TO_ALL   of leaf digests  -> rank r owns the contiguous leaves [r*M/world, (r+1)*M/world) and builds that subtree,
 *   ALL_GATHER   of the `world` subtree roots -> every rank finishes the top log2(world) levels.
 * The query phase locates leaves by value on every rank (ALL_REDUCE_MIN of the global index: first match wins,
 * quirk Q7) and assembles the Merkle paths from their owners (ALL_REDUCE_SUM over disjoint bytes).
 * The library calls `fn` at those points with the payload in the caller-provided device buffers:
 *   ALL_TO_ALL         send = world chunks of `bytes` (chunk p goes to rank p), recv = world chunks (chunk p from rank p)
 *   ALL_GATHER         send = `bytes`, recv = world * `bytes` in rank order
 *   ALL_REDUCE_MIN_U64 / ALL_REDUCE_SUM_U8   in place on send, `bytes` in total
 *   ALL_TO_ALL_SLICE   one slice of a commitment's digest all-to-all (large commitments are hashed in MS_SHARD_SLICES slices so that the digests of a
 *                      slice travel while the next one is hashed): peer p's piece is the `bytes` at offset + p * stride of BOTH buffers, with
 *                      (offset, stride) from ms_shard_slice_layout(ctx, ..) inside the callback
 * The library has synchronised its stream before the call; `fn` returns 0 once the result is visible to the device.
 * Commitments with fewer than MS_SHARD_MIN_LEAVES (env, default 32768) leaf groups stay replicated.
 * world must be a power of two; world = 1 switches sharding off.  ms_lde_read / ms_fri_round_codeword_read
 * are not available for sharded commitments (each rank holds its part only). */
typedef enum { MS_XCHG_ALL_TO_ALL = 0, MS_XCHG_ALL_GATHER = 1, MS_XCHG_ALL_REDUCE_MIN_U64 = 2, MS_XCHG_ALL_REDUCE_SUM_U8 = 3, MS_XCHG_ALL_TO_ALL_SLICE = 4,
               MS_XCHG_GATHER = 5 } ms_xchg_op;
typedef int (*ms_exchange_fn)(void* user, int op, size_t bytes);
int ms_set_shard(ms_ctx* ctx, int rank, int world, void* d_send, void* d_recv, size_t cap_bytes, ms_exchange_fn fn, void* user);
int ms_shard_slice_layout(ms_ctx* ctx, size_t* offset, size_t* stride);
/* The production form: the library runs the four collectives itself with RCCL (ncclSend/ncclRecv in one group, ncclAllGather,
 * ncclAllReduce) ON THE CONTEXT'S STREAM - stream-ordered with the kernels on both sides, no stream synchronisation, no host
 * callback - and owns the two exchange buffers (`cap_bytes` each: 32 * leaf groups of the largest commitment / world + 4 MiB).
 * Rank 0 obtains `unique_id` (ncclGetUniqueId) with ms_rccl_unique_id and hands the 128 bytes to every rank through whatever
 * channel the caller has (bench.py: a torch.distributed broadcast); every rank then calls ms_set_shard_rccl (collective:
 * ncclCommInitRank).  librccl.so is bound at run time (dlopen; env MS_RCCL_LIB overrides the name): MS_ERR_HIP if it is absent.
 * ms_set_shard above stays as the seam for callers without RCCL (the gloo tests on the kernel-emulation build). */
int ms_rccl_unique_id(uint8_t out[128]);
int ms_set_shard_rccl(ms_ctx* ctx, int rank, int world, const uint8_t unique_id[128], size_t cap_bytes);
/* (tests: with env MS_SHARD_WORLD1=1 at ms_create, ms_set_shard / ms_set_shard_rccl accept world = 1 WITH buffers / a unique id and run the sharded code paths on that one
 * rank - every exchange a transfer to itself - so that the sharded kernels and, through ms_set_shard_rccl, the RCCL calls execute through whole proofs on one GPU.) */
/* the four collectives on a ONE-rank RCCL communicator with known payloads: checks the run-time binding (symbols, enums, the
 * by-value ncclUniqueId) and the stream ordering on a single GPU */
int ms_rccl_selftest(ms_ctx* ctx);
/* collective calls [0..3] and bytes sent [4..7] by this rank so far, per ms_xchg_op (the gather to rank 0 is counted with the all-gathers) */
int ms_shard_stats(ms_ctx* ctx, uint64_t out[8]);
/* r04 - the coefficient-domain work of a sharded proof is partitioned too (env MS_SHARD_DIST=0: replicated as before): the raw-trace tree by contiguous
 * leaf ranges (every rank holds the trace; only the subtree roots travel), and for every FRI round whose commitment is sharded the round polynomial BY
 * COEFFICIENT RANGE - rank k holds the coefficients [k*S, (k+1)*S), S = domain / (blowup * world): the fold is local, the DEEP quotient's suffix sums take
 * the sum over the higher ranks as carry-in (ALL_GATHER of one aggregate per rank), even(z) / odd(z) and the DEEP-ALI values of ms_eval_ext are Horner
 * combinations of the ranks' partial sums (ALL_GATHER), the trimmed length rides on the subtree-root ALL_GATHER.  The query phase computes every quotient
 * polynomial by coefficient range as well and ALL_GATHERs the ranks' slices of the proof - or, with ms_shard_proof_on_root(ctx, 1), GATHERs them to rank 0 only
 * (MS_XCHG_GATHER: send = `bytes`, rank 0's recv = world * `bytes` in rank order): ms_fri_proof_size is then 0 on the other ranks.  INTT, the constraint
 * polynomials, the mix and the commitments below MS_SHARD_MIN_LEAVES stay replicated. */
int ms_shard_proof_on_root(ms_ctx* ctx, int on);
int ms_shard_proof_is_elsewhere(const ms_ctx* ctx); /* 1 on a rank != 0 whose finished proof was assembled on rank 0 (ms_shard_proof_on_root) */
/* 1 if FRI round `round` of the current proof is worked on by coefficient range (its polynomial distributed over the ranks), 0 if it is replicated */
int ms_shard_round_is_distributed(ms_ctx* ctx, int round);

/* ---- src/util.rs:4-44, src/starks.rs:268-332 (host-only config math) ----- */
int ms_is_power_of_two(uint64_t n);
long ms_logarithm_of_two_k(uint64_t n, uint64_t base); /* -1: not a power of 2, -2: not a power of base */
uint64_t ms_ceil_log2_k(uint64_t n, uint64_t base);
int ms_num_queries(ms_field f, uint64_t security_bits, uint64_t blowup, uint64_t steps,
                   uint64_t* linking_queries, uint64_t* fri_queries_per_round); /* starks.rs:312-332 */
uint64_t ms_root_of_unity(ms_field f, uint64_t n); /* Radix2EvaluationDomain::new(n).group_gen  (TraceTable::omega, air.rs:75) */

/* ---- Stark::prove stages (src/starks.rs:59-169) --------------------------- */
/* 1.1  MerkleTree::new(trace.get_data(), lpn) -> root.  starks.rs:68-73, air.rs:15-59.
 *      `trace` is the N x w row-major matrix (N a power of two).  Also uploads the trace. */
int ms_trace_commit(ms_ctx* ctx, const uint64_t* trace_rowmajor, size_t N, size_t w, size_t lpn, uint8_t root[32]);
/*      Same, for a trace already resident in HBM (device pointer, same layout).  Elements must be canonical (< p) like the
 *      host path's; the range check runs inside the transposing kernel and the call returns MS_ERR_ARG if any element is >= p. */
int ms_trace_commit_device(ms_ctx* ctx, const void* d_trace_rowmajor, size_t N, size_t w, size_t lpn, uint8_t root[32]);
/* Optional prefetch of the NEXT proof's trace (r05): queues the copy of a page-locked (ms_pinned_alloc) row-major N x w trace into the device buffer the proof in
 * flight does not use, on an SDMA engine, and returns at once; the ms_trace_commit that later names the same pointer and shape finds the trace on the device instead
 * of waiting for the transfer.  Callable any time after the previous ms_trace_commit has returned - typically right behind it, so that the ~24 MiB travel while
 * the current proof computes.  A hint: pageable memory, MS_UPLOAD=hip or a busy engine queue nothing (MS_OK all the same) and ms_trace_commit uploads as before.
 * `trace` must stay valid and unchanged until that ms_trace_commit returns; one prefetch in flight per context (a second one waits for the first). */
int ms_trace_upload_async(ms_ctx* ctx, const uint64_t* trace_rowmajor, size_t N, size_t w);
/* 1.2a TraceTable::get_trace_polys: per-column INTT.  air.rs:147-160. */
int ms_interpolate(ms_ctx* ctx);
/* 1.2b constraint polynomial appended as sum_t scalars[t] * poly[idx[t]] (the
 *      closures of tests/e2e_goldilocks.rs:48-59 are such combinations), or
 *      uploaded verbatim (N canonical coefficients).  air.rs:127-144. */
int ms_polys_lincomb(ms_ctx* ctx, const uint64_t* scalars, const int* idx, int k);
int ms_polys_append(ms_ctx* ctx, const uint64_t* coeffs, size_t n);
int ms_polys_count(const ms_ctx* ctx);
int ms_poly_read(ms_ctx* ctx, int i, uint64_t* out /* N */);
/* 1.2c coset LDE of every constraint polynomial over Radix2(blowup*N).get_coset(shift)
 *      and MerkleTree::new over the L x c row-major LDE matrix.  starks.rs:80-95. */
int ms_lde_commit(ms_ctx* ctx, size_t blowup, uint64_t shift, size_t lpn, uint8_t root[32]);
int ms_lde_read(ms_ctx* ctx, uint64_t* out_rowmajor /* L*c */);
/* 1.3  validity = sum_i r^i f_i (remainder of divide_by_vanishing_poly; quirk Q1).  starks.rs:108-119. */
int ms_mix(ms_ctx* ctx, uint64_t r);
int ms_validity_read(ms_ctx* ctx, uint64_t* out /* ms_validity_len(ctx): N, or 2N after ms_mix_cubic */);
/* BUILD-DEFINED, no reference counterpart (BASELINE configs[4] "degree-3 constraints"; the reference cannot express them: its validity polynomial is the
 * remainder of divide_by_vanishing_poly, starks.rs:118-119, quirk Q1).  In place of ms_mix, with the TRUE quotient:
 *   C_t(x) = P_j(w x) - P_a(x) P_b(x) P_c(x) - s_t P_d(x)         spec[t] = {j, a, b, c, d} (polynomial indices), w = the trace domain's generator
 *   validity(x) = (sum_t r^t C_t(x)) (x - w^(N-1)) / (x^N - 1)     2N coefficients; MS_ERR_SHAPE if the division is not exact (a row 0..N-2 violates a constraint)
 * Evaluated on the LDE domain of the preceding ms_lde_commit (blowup >= 4), interpolated back.  ms_eval_ext / ms_validity_read / ms_fri_begin then work on the
 * 2N-coefficient validity polynomial.  Checked against a big-integer restatement in the tests; self-verified at full size. */
int ms_mix_cubic(ms_ctx* ctx, uint64_t r, const int* spec /* [ncons][5] */, const uint64_t* s /* [ncons] */, int ncons);
size_t ms_validity_len(const ms_ctx* ctx);
/* 2.   DEEP-ALI: out[t][i] = f_i(z_t) for the c constraint polys, out[t][c] = validity(z_t);
 *      z: q*E limbs, out: q*(c+1)*E limbs.  starks.rs:124-151, field.rs:23-32. */
int ms_eval_ext(ms_ctx* ctx, const uint64_t* z, int q, uint64_t* out);

/* ---- Fri::prove stages (src/fri.rs:53-189) -------------------------------- */
/* commit phase, round 0: FriRound::new(extend(validity), (deg+1)*blowup).  fri.rs:73-82, 314-352.
 * (root0 is returned for inspection; the reference never writes it to the transcript.) */
int ms_fri_begin(ms_ctx* ctx, size_t blowup, size_t rounds, uint8_t root0[32]);
/* z -> B = [even(z), odd(z)] (2*E limbs).  fri.rs:89-94, 354-359. */
int ms_fri_deep(ms_ctx* ctx, const uint64_t* z, uint64_t* B);
/* alpha -> fold, DEEP quotient (folded - B(alpha))/(x - z), next FriRound, root.  fri.rs:96-109. */
int ms_fri_fold_commit(ms_ctx* ctx, const uint64_t* alpha, uint8_t root[32]);
int ms_fri_round_info(ms_ctx* ctx, int round, uint64_t* ncoef, uint64_t* domain_size);
int ms_fri_round_poly_read(ms_ctx* ctx, int round, uint64_t* out /* ncoef*E */);
int ms_fri_round_codeword_read(ms_ctx* ctx, int round, uint64_t* out /* D*E */);
/* query phase for `nq` betas (already converted from challenge bytes to u64, fri.rs:121-126).
 * Builds the serialised FriProof ("MSFP" layout below) in HBM.  fri.rs:115-189. */
int ms_fri_query(ms_ctx* ctx, const uint64_t* betas, int nq);
size_t ms_fri_proof_size(const ms_ctx* ctx);
int ms_fri_proof_read(ms_ctx* ctx, uint8_t* out);
/* The same read-back without blocking: the copy runs on the context's own copy stream behind the query phase and `out` (page-locked:
 * ms_pinned_alloc) is complete once ms_fri_proof_wait returns; the next proof's stages may be issued meanwhile - the next
 * ms_fri_query orders itself behind the copy on the device.  One read-back in flight per context. */
int ms_fri_proof_read_async(ms_ctx* ctx, uint8_t* out);
int ms_fri_proof_wait(ms_ctx* ctx);
/* how the last read-back of this context travelled: 1 = an SDMA copy engine through the HSA runtime (hsa_amd_memory_async_copy_on_engine, forced onto the engine:
 * never a blit kernel; the default for ms_fri_proof_read_async when the HSA runtime binds), 0 = the HIP runtime's copy (hipMemcpyAsync; env MS_READBACK=hip, the
 * blocking ms_fri_proof_read, or the fall-back when the engine refused the copy). */
int ms_io_engine(const ms_ctx* ctx);
/* which HSA runtime the copy engines were bound through: the path of the libhsa-runtime64 ALREADY mapped into the process (the one HIP itself runs on; the library
 * never loads a runtime of its own), "" while none is bound.  A profiled or otherwise mixed-runtime run can be recognised in its records by this. */
const char* ms_io_runtime_path(void);
/* Failure semantics of the engine copies (upload and read-back alike): an engine that refuses a copy -> this context uses the HIP runtime's copies from then on; a copy
 * the engine reports as FAILED -> done again through the HIP runtime, the call succeeds if that does; a copy that does not COMPLETE within 20 s (env
 * MS_SDMA_TIMEOUT_S) -> MS_ERR_HIP, and the context is POISONED: the engine may still be accessing the caller's buffer and the device blob, so every later entry
 * point returns MS_ERR_STATE (ms_last_error says why), nothing is reused, and only ms_destroy is valid - it waits for the transfer without limit, then frees.
 * The caller must keep its buffer alive until ms_destroy has returned. */
/* ms_fri_query with the FriProof written WHERE IT IS READ: the query-phase kernels store the MSFP blob straight into `out` - page-locked
 * host memory from ms_pinned_alloc (mapped into the device's address space) or device memory - of `cap` bytes; no read-back copy.  The
 * blob is complete when the call returns.  *len receives the blob size; with cap smaller than that nothing is computed and MS_ERR_ARG is
 * returned (out = NULL, cap = 0: size query, MS_OK).  One buffer per proof in flight on the caller's side: the library keeps no reference to
 * `out` after the call.  ms_fri_proof_read* do not apply to such a proof (MS_ERR_STATE). */
int ms_fri_query_into(ms_ctx* ctx, const uint64_t* betas, int nq, uint8_t* out, size_t cap, size_t* len);
/* MSFP layout — for each window (previous, round) in order, for each beta in order:
 *     6*E u64   x1 y1 x2 y2 x3 y3                      (FriProof.points,    fri.rs:148-154)
 *     u64 qlen, qlen*E u64 quotient coefficients       (FriProof.quotients, fri.rs:159-167)
 *     MerklePath(y1), MerklePath(y2)                   (FriProof.queries,   fri.rs:170-172)
 * MerklePath = u64 leaf_index | lpn*E u64 leaf_neighbours | u64 nlevels | nlevels*ic*32 bytes
 *              (merkle.rs:272-298; leaf located BY VALUE, first match, quirk Q7). */

/* ---- Tree trait (src/merkle.rs:8-30) on its own -------------------------- */
/* MerkleTree::new over `leaf_num` elements of `ext` limbs each; writes all nodes
 * (level-major, root last) if nodes_out != NULL.  ext in {1, E}. */
int ms_merkle_commit(ms_ctx* ctx, const uint64_t* leafs, size_t leaf_num, int ext, size_t lpn, size_t ic,
                     uint8_t* nodes_out, size_t nodes_cap, size_t* nnodes, uint8_t root[32]);

/* MerkleTree::generate_proof(&leaf) (src/merkle.rs:272-288): builds the tree, locates `leaf` (ext limbs) BY VALUE,
 * first match (merkle.rs:216-225), and writes the serialised MerklePath (layout above) to path_out.
 * inner_children is 2 (get_parent_idx, merkle.rs:203, is only right for binary trees — DESIGN.md quirk Q14).
 * *path_len receives the path size; MS_ERR_LEAF_NOT_FOUND if the value is not a leaf. */
int ms_merkle_prove(ms_ctx* ctx, const uint64_t* leafs, size_t leaf_num, int ext, size_t lpn, const uint64_t* leaf,
                    uint8_t* path_out, size_t cap, size_t* path_len);

/* ---- standalone transforms (NTT micro-benchmark + parity tests) ---------- */
/* `batch` vectors of n elements each, contiguous; natural order in and out. */
int ms_ntt(ms_ctx* ctx, uint64_t* data, size_t n, size_t batch, int inverse);
/* out[b][i] = P_b(shift * g_L^i): coeffs [batch][ncoef] -> out [batch][L]. */
int ms_coset_lde(ms_ctx* ctx, const uint64_t* coeffs, size_t ncoef, size_t batch, uint64_t shift, uint64_t* out, size_t L);
/* device-resident benchmark entry: runs the LDE stage of the current session again
 * (no host copies); used by bench.py to time the NTT kernels in isolation. */
int ms_bench_lde(ms_ctx* ctx, size_t blowup, uint64_t shift);
/* measurement aid: between begin/end every kernel launch is bracketed by HIP events on the
 * launching stream; end() writes a JSON object {kernel: {launches, ms, alg_bytes}} (build-defined,
 * no reference counterpart). */
/* diagnostic: out[i] = a[i] (op) b[i] computed ON THE DEVICE with the arithmetic class the NTT tiles use (Goldilocks: the exec-masked
 * inline-asm class GLM, whose gfx950 wait states are managed by hand and which no CPU build can execute; BabyBear: BB) - so that a
 * test can compare every operation with big-integer arithmetic on directed edge values.  a, b canonical.  op: 0 add, 1 sub, 2 mul,
 * 3 mul by the table form of b (mul_tw(a, to_tw(b))), 4 a * 2^32, 5 a * 2^64, 6 a * 2^(b mod 96) through the compile-time shift
 * chains of the butterflies (Goldilocks only), 7 fold: a + (b mod 2^31) * 2^64 mod p (Goldilocks only).  No reference counterpart. */
int ms_arith_selftest(ms_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int ms_profile_begin(ms_ctx* ctx);
int ms_profile_end(ms_ctx* ctx, char* json_out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
