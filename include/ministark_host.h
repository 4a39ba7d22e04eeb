/* ministark_host.h — C entry points of libministark_host.so: the host-side mirror of the reference's prover API ABOVE
 * the C ABI of ministark.h (mini-stark_amd/host/stark_host.cpp: StarkConfig::new src/starks.rs:268-310, Stark::prove
 * src/starks.rs:59-169, Stark::verify src/starks.rs:171-235 with Fri::verify src/fri.rs:191-290 and MerkleRoot::check_proof
 * src/merkle.rs:312-338) and the whole-proof wire format.  The reference is Rust and derives no serialisation for StarkProof
 * (src/starks.rs:21-28), FriProof (src/fri.rs:17-22) or MerklePath (src/merkle.rs:293-298): both layouts below are build-defined.
 *
 * MSSP v1 (little-endian):
 *   u32 magic 'MSSP' | u32 version 1 | u32 E | u32 c | u32 q | u32 rounds | u64 len(arthur) | u64 len(fri blob)
 *   trace_commit[32] | constrain_trace_commit[32]
 *   constrain_queries q*c*E u64 | validity_queries q*E u64                     (StarkProof.constrain_queries / validity_query)
 *   fri_roots rounds*32   (round 0 first; not in the reference's StarkProof: round 0's root never reaches its transcript)
 *   arthur bytes          (the prover's transcript, StarkProof.arthur; build-defined SHA-256 chain, NOT nimue's bytes)
 *   FriProof in the MSFP layout of ministark.h
 * The Python mirror writes the same bytes (mini_stark_amd.stark.StarkProof.to_bytes). */
#ifndef MINISTARK_HOST_H
#define MINISTARK_HOST_H
#include <stddef.h>
#include <stdint.h>
#include "ministark.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct msh_stark msh_stark;
typedef struct {
  uint32_t e, c, q, rounds;
  uint64_t arthur_len, fri_blob_len;
  const uint8_t *trace_commit, *constrain_trace_commit, *fri_roots, *arthur, *fri_blob;
  const uint64_t *constrain_queries, *validity_queries;
} msh_proof_view;
/* one MerklePath / one (window, query) record of the MSFP blob (ministark.h): views INTO the parsed buffer */
typedef struct { uint64_t leaf_index, nlevels; const uint64_t* leaf_neighbours; /* 2*E */ const uint8_t* levels; /* nlevels * 2 * 32 */ } msh_merkle_path_view;
typedef struct { const uint64_t* points; /* 6*E: x1 y1 x2 y2 x3 y3 */ uint64_t qlen; const uint64_t* quotient; /* qlen*E */ msh_merkle_path_view path[2]; } msh_fri_query_view;

/* StarkConfig::new + Stark::new (starks.rs:268-310, 40-57); NULL and *err = MS_ERR_SHAPE for < 20 security bits (starks.rs:317-320) */
msh_stark* msh_stark_new(ms_ctx* ctx, int field, uint64_t security_bits, uint64_t blowup, uint64_t steps, uint64_t trace_columns, int* err);
void msh_stark_free(msh_stark* h);
int msh_stark_config(const msh_stark* h, uint64_t* rounds, uint64_t* constrain_queries, uint64_t* fri_queries);
/* Stark::prove (starks.rs:59-169): trace from host memory or already in HBM; transitions as lincombs of trace polynomials */
int msh_stark_prove(msh_stark* h, const uint64_t* trace_host, const void* trace_dev, size_t N, size_t w, int ntrans, const int* tr_k,
                    const uint64_t* tr_scalars, const int* tr_idx, int read_fri_proof);
/* A pipelined caller names the page-locked (ms_pinned_alloc) row-major trace of the proof AFTER the next msh_stark_prove: that call hands it to
 * ms_trace_upload_async right behind its own trace commitment, so the upload overlaps the proof (one-shot: name it again before every prove). */
void msh_stark_next_trace(msh_stark* h, const uint64_t* trace_host, size_t N, size_t w);
size_t msh_proof_arthur(const msh_stark* h, uint8_t* out, size_t cap);
int msh_proof_commits(const msh_stark* h, uint8_t* trace_commit, uint8_t* lde_commit);
size_t msh_proof_evals(const msh_stark* h, uint64_t* out, size_t cap_elems);
size_t msh_proof_fri_roots(const msh_stark* h, uint8_t* out, size_t cap);
/* read_fri_proof of msh_stark_prove: 0 = the FRI proof stays in HBM, 1 = read back before returning, 2 = read back asynchronously
 * (ms_fri_proof_read_async: the bytes travel into the slot's page-locked buffer while the caller goes on, e.g. into the next
 * msh_stark_prove); msh_proof_wait - and every accessor below that touches the blob - waits for them; 3 = the query-phase kernels write
 * the blob straight into the slot's page-locked buffer (ms_fri_query_into), complete on return.
 * The mirror holds TWO proof slots: msh_stark_prove k + 1 reuses the slot of proof k - 1, so proof k stays whole (msh_prev_proof_*) while
 * k + 1 is computed - including a mode-2 blob that is still arriving. */
int msh_proof_wait(const msh_stark* h);
size_t msh_proof_fri_blob(const msh_stark* h, uint8_t* out, size_t cap);
size_t msh_proof_challenges(const msh_stark* h, uint64_t* out, size_t cap_elems);
size_t msh_proof_num_polys(const msh_stark* h);
size_t msh_prev_proof_arthur(const msh_stark* h, uint8_t* out, size_t cap);
size_t msh_prev_proof_fri_roots(const msh_stark* h, uint8_t* out, size_t cap);
size_t msh_prev_proof_fri_blob(const msh_stark* h, uint8_t* out, size_t cap);
/* FNV-1a (64-bit words) over the FRI blob of the last (which = 0) / previous (which = 1) proof, read in place from its page-locked slot */
uint64_t msh_proof_blob_checksum(const msh_stark* h, int which);
/* the same over one 64-bit word per `stride_bytes` plus the last word */
uint64_t msh_proof_blob_sample(const msh_stark* h, int which, size_t stride_bytes);
/* Stark::verify (starks.rs:171-235) on the CPU.  A PARITY MIRROR of the reference's verifier, including what it does NOT bind
 * (INTEGRATION.md "verifier"): 1 accepted, 0 rejected, < 0 malformed. */
int msh_stark_verify(const msh_stark* h, const uint64_t* constrains, size_t c, size_t N, const uint8_t* arthur, size_t arthur_len, const uint8_t* trace_commit,
                     const uint8_t* lde_commit, const uint64_t* evals, size_t nevals, const uint8_t* fri_roots, size_t nroots, const uint8_t* blob, size_t blob_len,
                     int zero_display_empty, char* why, size_t why_cap);
/* MSSP: serialise the last proof (returns the size needed; writes when cap suffices), parse (views into the buffer), verify from bytes */
size_t msh_proof_serialize(const msh_stark* h, uint8_t* out, size_t cap);
int msh_proof_parse(const uint8_t* data, size_t len, msh_proof_view* out);
int msh_stark_verify_mssp(const msh_stark* h, const uint64_t* constrains, size_t c, size_t N, const uint8_t* data, size_t len, int zero_display_empty, char* why, size_t why_cap);
/* FriProof (fri.rs:17-22) from its MSFP bytes: the compiled twin of the Rust shim's FriProof::from_msfp.  windows = rounds - 1; returns the
 * number of (window, query) records (windows * nq; the first `cap` are written to `out`) or -1 if the bytes do not parse exactly. */
int msh_fri_proof_parse(const uint8_t* blob, size_t len, uint32_t e, uint32_t windows, uint32_t nq, msh_fri_query_view* out, size_t cap);
/* synthetic trace of the build-defined degree-3 wide AIR (ms_mix_cubic): col_j[i+1] = col_j[i] col_{j+1}[i] col_{j+2}[i] + s_j col_{j+3}[i]; length x w row-major, w scalars */
int msh_cubic_rows(uint64_t p, size_t length, size_t w, uint64_t seed, uint64_t* out, uint64_t* scalars);
/* synthetic Fibonacci-AIR trace of the benchmark workload (N x 3 row-major) */
int msh_fibonacci_rows(uint64_t p, size_t length, size_t steps, uint64_t secret_b, uint64_t pad_seed, uint64_t* out);
#ifdef __cplusplus
}
#endif
#endif
