/* prove_c_caller.c — a plain-C caller of the two shared libraries, sequencing the calls the way the Rust shim of INTEGRATION.md
 * would (no Rust toolchain exists in this image; this is the nearest compiled stand-in for it):
 *
 *   libministark.so       the accelerated path behind the C ABI of include/ministark.h (HIP, gfx950)
 *   libministark_host.so  StarkConfig::new / Stark::prove / Stark::verify above it (include/ministark_host.h)
 *
 * Flow = the reference's integration test (tests/e2e_goldilocks.rs:98-114): build the Fibonacci AIR trace, derive the verifier's
 * copy of the constraint polynomials, prove, serialise the proof (MSSP), verify it from the bytes, reject a tampered copy.
 *
 *   gcc -O2 -Iinclude examples/prove_c_caller.c -o prove_c_caller -Lmini-stark_amd -lministark -lministark_host -Wl,-rpath,$PWD/mini-stark_amd
 *   ./prove_c_caller [field 0|1] [log2 rows] [blowup]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ministark.h"
#include "ministark_host.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != MS_OK) { fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, ctx ? ms_last_error(ctx) : "-"); return 1; } } while (0)

int main(int argc, char** argv) {
  const int field = argc > 1 ? atoi(argv[1]) : 0;
  const int log_rows = argc > 2 ? atoi(argv[2]) : 10;
  const uint64_t blowup = argc > 3 ? (uint64_t)atoll(argv[3]) : 8;
  const uint64_t p = field == 0 ? 0xFFFFFFFF00000001ULL : 2013265921ULL;
  const size_t N = (size_t)1 << log_rows, steps = N - 1, w = 3, c = 6;     /* "2^k rows" = steps 2^k - 1 (one padding row) */
  ms_ctx* ctx = NULL;
  CHECK(ms_create(&ctx, 0, field == 0 ? MS_FIELD_GOLDILOCKS : MS_FIELD_BABYBEAR, MS_FLAGS_DEFAULT));

  /* Provable::trace(): the N x 3 Fibonacci matrix (tests/e2e_goldilocks.rs:20-63) */
  uint64_t* trace = (uint64_t*)malloc(N * w * 8);
  if (!trace || msh_fibonacci_rows(p, N, steps, 2, 0x5EED, trace)) { fprintf(stderr, "trace\n"); return 1; }
  /* the three transition closures as linear combinations of trace polynomials (tests/e2e_goldilocks.rs:48-59) */
  const uint64_t omega = ms_root_of_unity(field == 0 ? MS_FIELD_GOLDILOCKS : MS_FIELD_BABYBEAR, N);
  const int tr_k[3] = {2, 2, 3};
  const uint64_t tr_s[7] = {omega, p - 1, omega, p - 1, 1, p - 1, p - 1};
  const int tr_i[7] = {0, 1, 0, 1, 2, 0, 1};

  /* verifier's side: trace.derive_constrains() (tests/e2e_goldilocks.rs:101) -> c polynomials of N coefficients */
  uint8_t root[32];
  CHECK(ms_trace_commit(ctx, trace, N, w, c, root));
  CHECK(ms_interpolate(ctx));
  for (int i = 0, off = 0; i < 3; off += tr_k[i], i++) CHECK(ms_polys_lincomb(ctx, tr_s + off, tr_i + off, tr_k[i]));
  uint64_t* constrains = (uint64_t*)malloc(c * N * 8);
  for (size_t i = 0; i < c; i++) CHECK(ms_poly_read(ctx, (int)i, constrains + i * N));

  /* StarkConfig::new(20, blowup, steps, constrain_number) + Stark::new + Stark::prove */
  int err = 0;
  msh_stark* stark = msh_stark_new(ctx, field, 20, blowup, steps, c, &err);
  if (!stark) { fprintf(stderr, "msh_stark_new: %d\n", err); return 1; }
  CHECK(msh_stark_prove(stark, trace, NULL, N, w, 3, tr_k, tr_s, tr_i, 1));
  const size_t len = msh_proof_serialize(stark, NULL, 0);
  uint8_t* wire = (uint8_t*)malloc(len);
  if (!len || msh_proof_serialize(stark, wire, len) != len) { fprintf(stderr, "serialize\n"); return 1; }

  /* Stark::verify from the bytes; then a tampered copy (last byte: a Merkle sibling digest) must be rejected */
  char why[256];
  const int ok = msh_stark_verify_mssp(stark, constrains, c, N, wire, len, 1, why, sizeof why);
  wire[len - 1] ^= 1;
  const int bad = msh_stark_verify_mssp(stark, constrains, c, N, wire, len, 1, why, sizeof why);
  msh_proof_view v;
  wire[len - 1] ^= 1;
  if (msh_proof_parse(wire, len, &v)) { fprintf(stderr, "parse\n"); return 1; }
  printf("C-CALLER field %d rows 2^%d blowup %llu: proof %zu bytes (rounds %u, E %u, c %u, q %u), verify %s, tampered %s (%s), trace root %02x%02x%02x%02x\n",
         field, log_rows, (unsigned long long)blowup, len, v.rounds, v.e, v.c, v.q, ok == 1 ? "accepted" : "REJECTED", bad == 0 ? "rejected" : "ACCEPTED", why,
         v.trace_commit[0], v.trace_commit[1], v.trace_commit[2], v.trace_commit[3]);
  msh_stark_free(stark);
  ms_destroy(ctx);
  free(trace); free(constrains); free(wire);
  return (ok == 1 && bad == 0) ? 0 : 2;
}
