"""CPU-side checks of the boundary: the library builds for gfx950, loads, and exports every
symbol include/openintel_hip.h declares.  No compute calls (no GPU here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "openintel_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(oi_[a-z_0-9]+)\s*\(", hdr)))


def test_library_builds_and_exports_every_declared_symbol():
    from openintel_amd import build, _lib
    build.build()
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "missing export: " + name
    assert sorted(_lib.SIGNATURES) == declared, "python binding table out of sync with the header"
    assert lib.oi_abi_version() == 1


def test_library_is_gfx950_only_and_has_no_cpu_fallback():
    import ctypes as C
    import subprocess
    from openintel_amd import _lib
    lib = _lib.load()
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          "--input=" + _lib.LIB_PATH], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets
    import torch
    if not torch.cuda.is_available():
        h = C.c_void_p()
        rc = lib.oi_create(0, C.byref(h))
        assert rc != 0 and not h.value            # no device -> loud failure, never a CPU path
        assert b"no HIP device" in lib.oi_last_error() or lib.oi_last_error()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "openintel_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oi_oracle" not in src and "liboi_oracle" not in src, f


def test_product_build_has_no_ablation_switches():
    """A/B switches, 'wrong results' ablation variants and the v1 kernels are reachable only in a -DOI_ABLATION
    build (oi_internal.h: oi_ablation_env): their environment-variable names must not even be in the product
    binary, so a stray variable cannot change what the C ABI returns."""
    from openintel_amd import build, _lib
    build.build()
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"OI_KS_DEBUG", b"OI_CS_DEBUG", b"OI_BM25_SCAN_DBG", b"OI_HEADLINE_DBG", b"OI_COSINE_V1",
                 b"OI_SELECT_V1", b"OI_LEXICON_V1", b"OI_HEADLINE_V1", b"OI_NO_OVERLAP", b"OI_CHUNK_GROWTH",
                 b"OI_FIRST_CHUNK_MULT", b"OI_KS_SHAPE", b"OI_BF16_SOLO", b"OI_HEADLINE_TILE", b"OI_HEADLINE_TIMING",
                 b"OI_LEXICON_V2", b"OI_LEX_DBG", b"OI_LEX_GRID", b"OI_SCREEN_CUS", b"OI_QUAD_DBG", b"OI_BF16_NO_QUAD",
                 b"OI_BM25_WAVE_DBG", b"OI_BM25_WAVE_TIMING", b"OI_BM25_WAVE_WGS", b"OI_SEG_DBG",
                 b"OI_BM25_STREAM_W", b"OI_BM25_STREAM_CHUNKS", b"OI_BM25_STREAM_WGS", b"OI_BM25_STREAM_TIMING", b"OI_BM25_FIRST_DIV"):
        assert name not in blob, name
    assert b"OI_COSINE_MODE" in blob and b"OI_BM25_MODE" in blob   # the two documented mode selectors stay


def test_null_handles_are_refused_without_touching_a_device():
    """Argument checks come before any HIP or RCCL call: a null handle is OI_ERR_INVALID_ARG with a message, on a box with no
    GPU too (round 3's new entry points included)."""
    import ctypes as C
    from openintel_amd import _lib
    lib = _lib.load()
    INVALID = _lib.OI_ERR_INVALID_ARG
    none = C.c_void_p(None)
    out = C.c_void_p()
    assert lib.oi_set_graph_replay(none, 1) == INVALID and lib.oi_last_error()
    assert lib.oi_comm_unique_id(none) == INVALID
    assert lib.oi_comm_create(none, none, 0, 1, C.byref(out)) == INVALID and not out.value
    assert lib.oi_index_finalize_sharded(none, none) == INVALID
    assert lib.oi_search_sharded(none, none, none, none, none, 1, 1, 1, _lib.OI_HOST, none, none, none) == INVALID
    lib.oi_comm_destroy(none)       # destroying nothing is a no-op, as for oi_destroy / oi_index_destroy
    lib.oi_destroy(none)
    lib.oi_index_destroy(none)
    assert lib.oi_set_stream(none, none) == INVALID and lib.oi_synchronize(none) == INVALID
    # the batch callers' entry points and the dip rows' scan
    assert lib.oi_social_summary_segmented(none, none, none, none, 0, none, 1, 0.2, _lib.OI_HOST, none) == INVALID
    assert lib.oi_lexicon_scan_segments_device(none, none, none, 0, 0, none, none, 1, 0.2, none, none, none) == INVALID
    assert lib.oi_headline_scan_rows(none, none, none, 0, none, 1, none, none, none, none, none, none, none, none) == INVALID
    assert b"null ctx" in lib.oi_last_error()


def test_a_host_without_rccl_gets_unsupported_not_a_crash():
    """the RCCL-missing path used to call dlerror() twice (the second call returns NULL: std::string(NULL) inside
    call_once aborts the process).  OI_RCCL_LIB points the loader at a library that does not exist; oi_comm_unique_id must come
    back with OI_ERR_UNSUPPORTED and the loader's message.  (A subprocess: the load is a process-wide one-shot.)"""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "from openintel_amd import _lib\n"
        "lib = _lib.load()\n"
        "buf = (C.c_uint8 * 128)()\n"
        "rc = lib.oi_comm_unique_id(buf)\n"
        "msg = lib.oi_last_error() or b''\n"
        "print(rc, msg.decode(errors='replace'))\n"
        "rc2 = lib.oi_comm_unique_id(buf)\n"          # the one-shot stays failed, still no crash
        "sys.exit(0 if rc == _lib.OI_ERR_UNSUPPORTED and rc2 == rc and b'RCCL is not usable' in msg and b'no_such_rccl' in msg else 3)\n")
    env = dict(os.environ, OI_RCCL_LIB="/nonexistent/libno_such_rccl.so", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-2000:])
