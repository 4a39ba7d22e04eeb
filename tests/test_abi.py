"""The C-ABI library loads (no GPU needed to dlopen) and exports every symbol include/ministark.h declares."""
import ctypes
import os
import re

import mini_stark_amd as ms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ministark.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ms_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    syms = declared_symbols()
    assert len(syms) >= 30
    path = ms.build_library()  # hipcc cross-compiles for gfx950 without a GPU (no-op when up to date)
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_emulation_build_exports_same_abi():
    emu = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libministark_emu.so"))
    assert not [s for s in declared_symbols() if not hasattr(emu, s)]


def test_no_fallback_when_library_missing(tmp_path):
    import pytest
    with pytest.raises(ms.MsError):
        ms.Context(ms.GOLDILOCKS, lib_path=str(tmp_path / "nope.so"))


def test_host_library_exports_its_header():
    """libministark_host.so (the host mirror above the C ABI) exports every msh_* function include/ministark_host.h declares."""
    from mini_stark_amd.host import build_host_library
    text = open(os.path.join(ROOT, "include", "ministark_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    syms = sorted(set(re.findall(r"\b(msh_[a-z0-9_]+)\s*\(", text)))
    assert len(syms) >= 15
    ctypes.CDLL(ms.build_library(), mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(build_host_library())
    assert not [s_ for s_ in syms if not hasattr(lib, s_)]


def test_rust_ffi_block_matches_header():
    """examples/rust_shim/src/ffi.rs (the reference-side binding; not compilable here: no cargo) declares exactly the functions
    include/ministark.h declares - no symbol missing, none invented - and every stage method of the safe wrapper calls one of them."""
    ffi = open(os.path.join(ROOT, "examples", "rust_shim", "src", "ffi.rs")).read()
    rust = sorted(set(re.findall(r"pub fn (ms_[a-z0-9_]+)\s*\(", ffi)))
    assert rust == declared_symbols()
    lib = open(os.path.join(ROOT, "examples", "rust_shim", "src", "lib.rs")).read()
    used = set(re.findall(r"\b(ms_[a-z0-9_]+)\s*\(", lib))
    assert used and used <= set(rust)
    for stage in ("ms_trace_commit", "ms_interpolate", "ms_polys_lincomb", "ms_lde_commit", "ms_mix", "ms_eval_ext", "ms_fri_begin", "ms_fri_deep",
                  "ms_fri_fold_commit", "ms_fri_query", "ms_fri_proof_read"):
        assert stage in used, stage
