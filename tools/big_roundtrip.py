#!/usr/bin/env python3
"""prove on the GPU -> Stark::verify on the CPU at sizes the oracle does not reach:  python tools/big_roundtrip.py LOG_ROWS
"accepted" = the parity mirror of the REFERENCE's verifier accepts (a round-trip check of the prover's plumbing, not a soundness
statement: the reference's FRI verifier leaves round 0's root, the window chaining and the last round unchecked)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mini_stark_amd as ms
from mini_stark_amd.host import HostStark, build_host_library
from mini_stark_amd.stark import fibonacci_air
build_host_library()
log_n = int(sys.argv[1])
ctx = ms.Context(ms.GOLDILOCKS)
steps = (1 << log_n) - 1
tt = fibonacci_air(ctx, steps)
hs = HostStark(ctx, 20, 8, steps, tt.constrain_number())
t = time.time(); constrains = hs.derive_constrains(tt); t_dc = time.time() - t
t = time.time(); proof = hs.prove(tt); t_pr = time.time() - t
t = time.time(); ok = hs.verify(constrains, proof); t_vf = time.time() - t
print(f"2^{log_n} rows: prove {t_pr:.2f} s (incl. {len(proof.fri_proof.blob) / 2**30:.2f} GiB FRI proof read-back), CPU verify {t_vf:.2f} s -> {'accepted' if ok else 'REJECTED: ' + hs.last_verify_error}", flush=True)
blob = bytearray(proof.fri_proof.blob); blob[16] ^= 1
proof.fri_proof = type(proof.fri_proof)(bytes(blob), device_resident=False)
print("tampered:", "accepted (BAD)" if hs.verify(constrains, proof) else "rejected: " + hs.last_verify_error)
