"""Fresh-process verifier for tests/test_gpu_parity.py::test_mssp_roundtrip_across_processes: reads an MSSP proof file written by
another process, re-derives the verifier's copy of the AIR constraints and runs Stark::verify from the bytes alone."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mini_stark_amd as ms  # noqa: E402
from mini_stark_amd.host import HostStark, build_host_library  # noqa: E402
from mini_stark_amd.stark import fibonacci_air  # noqa: E402

path, field, steps, blowup = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lib = sys.argv[5] if len(sys.argv) > 5 else None
build_host_library()
ctx = ms.Context(field, lib_path=lib)
tt = fibonacci_air(ctx, steps)
hs = HostStark(ctx, 20, blowup, steps, tt.constrain_number())
constrains = hs.derive_constrains(tt)          # trace.derive_constrains() on the verifier's side (tests/e2e_goldilocks.rs:101)
wire = open(path, "rb").read()
ok = hs.verify_bytes(constrains, wire)
bad = bytearray(wire); bad[-1] ^= 1   # the last byte is a Merkle sibling digest (the shipped quotient coefficients in the middle are only degree-bounded by the reference's verifier)
rejected = not hs.verify_bytes(constrains, bytes(bad))
print("VERIFY", "accepted" if ok else "REJECTED: " + hs.last_verify_error, "tampered-rejected" if rejected else "TAMPERED-ACCEPTED", len(wire), flush=True)
sys.exit(0 if ok and rejected else 1)
