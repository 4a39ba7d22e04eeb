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


def _c_arity():
    """name -> number of parameters, from the C declarations of include/ministark.h"""
    text = open(os.path.join(ROOT, "include", "ministark.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(ms_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else _split_top(args)
    return out


def _split_top(s):
    depth, n = 0, 1
    for ch in s:
        depth += ch in "([{<"
        depth -= ch in ")]}>"
        if ch == "," and depth == 0:
            n += 1
    return n


def _calls(text, prefix):
    """(name, number of top-level arguments) of every call `prefix...(` in Rust source `text`"""
    out = []
    for m in re.finditer(r"\b(%s[A-Za-z0-9_]*)\s*\(" % prefix, text):
        i, depth = m.end(), 1
        while depth and i < len(text):
            depth += text[i] in "([{"
            depth -= text[i] in ")]}"
            i += 1
        args = text[m.end():i - 1].strip()
        out.append((m.group(1), 0 if not args else _split_top(args)))
    return out


def test_rust_shim_files_call_the_abi_with_the_right_arity():
    """SURVEY 8(f) rank 4 as FILES: tree.rs (`impl Tree for GpuMerkleTree`), prove.rs (`Stark::prove_gpu`), fri_proof.rs
    (`FriProof::from_msfp`) exist without `unimplemented!()`; every `ms_*` call of the crate names a function of include/ministark.h with the
    header's number of arguments (declarations in ffi.rs and calls in lib.rs alike); every `gpu.<stage>()` call of tree.rs / prove.rs is a
    method lib.rs defines, with the right number of arguments.  (Not compiled here: no cargo; the C twin of from_msfp, msh_fri_proof_parse,
    is compiled and tested in tests/test_host_mirror.py.)"""
    src = os.path.join(ROOT, "examples", "rust_shim", "src")
    files = {n: open(os.path.join(src, n)).read() for n in ("ffi.rs", "lib.rs", "tree.rs", "prove.rs", "fri_proof.rs", "convert.rs")}
    code = {n: re.sub(r"//[^\n]*", "", t) for n, t in files.items()}      # comments out
    for n in ("tree.rs", "prove.rs", "fri_proof.rs", "lib.rs"):
        assert "unimplemented!" not in code[n] and "todo!" not in code[n], n
    assert "impl<F: FftField> Tree for GpuMerkleTree<Sha256, F>" in code["tree.rs"]
    assert "pub fn prove_gpu" in code["prove.rs"] and "pub fn from_msfp" in code["fri_proof.rs"]
    arity = _c_arity()
    # declarations: `pub fn ms_x(a: T, b: U) -> c_int;`
    for name, n in _calls(re.sub(r"pub fn ", "", code["ffi.rs"].split('extern "C" {', 1)[1]), "ms_"):
        assert arity.get(name) == n, (name, n, arity.get(name))
    # calls in the safe wrapper
    calls = _calls(code["lib.rs"], "ms_")
    assert len(calls) >= 15
    for name, n in calls:
        assert name in arity and arity[name] == n, ("lib.rs", name, n, arity.get(name))
    # gpu.<method>(...) in tree.rs / prove.rs against the methods of `impl Gpu`
    methods = {}
    for m in re.finditer(r"pub fn ([a-z_0-9]+)\s*\(\s*&(?:mut )?self\s*,?([^)]*)\)", code["lib.rs"]):
        rest = m.group(2).strip()
        methods[m.group(1)] = 0 if not rest else _split_top(rest)
    used = []
    for n in ("tree.rs", "prove.rs"):
        used += [(n,) + c for c in _calls(code[n], r"gpu\.")]
    assert len(used) >= 10
    for fname, call, n in used:
        meth = call.split(".", 1)[1]
        assert meth in methods and methods[meth] == n, (fname, call, n, methods.get(meth))
    for stage in ("trace_commit", "interpolate", "polys_lincomb", "lde_commit", "mix", "eval_ext", "fri_begin", "fri_deep", "fri_fold_commit", "fri_query"):
        assert any(c[1] == "gpu." + stage for c in used if c[0] == "prove.rs"), stage
    assert any(c[1] == "gpu.merkle_commit" for c in used if c[0] == "tree.rs")


def test_rust_shim_files_lex_as_rust_with_balanced_brackets():
    """No Rust toolchain in this image (SURVEY 8(f) rank 4 stays "uncompiled"); the least a file can be asked without one: it lexes as Rust without an error token
    (pygments' RustLexer) and its brackets balance.  Catches the truncated file / stray character class of damage, nothing more."""
    from pygments.lexers import RustLexer
    from pygments.token import Error, Punctuation
    src = os.path.join(ROOT, "examples", "rust_shim", "src")
    pairs = {")": "(", "]": "[", "}": "{"}
    for name in ("ffi.rs", "lib.rs", "convert.rs", "tree.rs", "prove.rs", "fri_proof.rs"):
        toks = list(RustLexer().get_tokens(open(os.path.join(src, name)).read()))
        assert len(toks) > 200 and not [v for t, v in toks if t is Error], name
        stack = []
        for t, v in toks:
            if t in Punctuation:
                for ch in v:
                    if ch in "([{":
                        stack.append(ch)
                    elif ch in ")]}":
                        assert stack and stack.pop() == pairs[ch], (name, ch)
        assert not stack, name
