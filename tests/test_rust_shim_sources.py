"""The Rust shim (integration/rust/) cannot be compiled in this image (no cargo/rustc), so the one thing that can rot
silently -- its `extern "C"` block drifting from include/openintel_hip.h -- is checked textually: every function the
header declares is bound in src/ffi.rs, nothing else is, and the argument counts agree."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c_decls():
    hdr = open(os.path.join(ROOT, "include", "openintel_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(oi_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def _rust_decls():
    src = open(os.path.join(ROOT, "integration", "rust", "src", "ffi.rs")).read()
    src = re.sub(r"//.*", "", src)
    block = src[src.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (oi_[a-z_0-9]+)\s*\(([^)]*)\)", block, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out


def test_rust_ffi_block_matches_the_header():
    c, r = _c_decls(), _rust_decls()
    assert len(c) >= 30
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in c:
        assert c[name] == r[name], (name, c[name], r[name])


def test_crate_layout():
    for f in ("Cargo.toml", "build.rs", "src/ffi.rs", "src/lib.rs"):
        assert os.path.exists(os.path.join(ROOT, "integration", "rust", f)), f
    lib = open(os.path.join(ROOT, "integration", "rust", "src", "lib.rs")).read()
    assert "impl PostAnalyzer for HipLexiconAnalyzer" in lib and "spawn_blocking" in lib
