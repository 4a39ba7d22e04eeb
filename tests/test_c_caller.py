"""A compiled C caller (examples/prove_c_caller.c) of libministark + libministark_host: the nearest checkable stand-in for the Rust
shim crate of SURVEY 8(f) rank 4 (no cargo / rustc in this image).  Here: linked against the kernel-EMULATION build (CPU); the GPU
suite links the same source against the HIP library (tests/test_gpu_parity.py::test_c_caller_on_gpu)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_and_run(libdir, libname, args, tmp_path):
    from mini_stark_amd.host import build_host_library, host_library_path
    build_host_library()
    exe = str(tmp_path / "prove_c_caller")
    hostdir = os.path.dirname(host_library_path())
    cmd = ["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "prove_c_caller.c"), "-o", exe,
           "-L", libdir, "-l" + libname, "-L", hostdir, "-lministark_host", "-Wl,-rpath," + libdir, "-Wl,-rpath," + hostdir]
    subprocess.check_call(cmd)
    return subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("field,log_rows,blowup", [(0, 6, 8), (1, 5, 4), (0, 4, 2)])
def test_c_caller_on_emulation(field, log_rows, blowup, tmp_path):
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-C", emu], stdout=subprocess.DEVNULL)
    out = build_and_run(emu, "ministark_emu", [field, log_rows, blowup], tmp_path)
    assert out.returncode == 0 and "verify accepted, tampered rejected" in out.stdout, (out.stdout, out.stderr)
