"""ctypes binding of libopenintel_hip.so (include/openintel_hip.h).

There is no CPU fallback: if the library is missing or no gfx950 device is visible the
calls raise.  `python -m openintel_amd.build` (or `__graft_entry__.build()`) compiles it.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libopenintel_hip.so")
# (No environment override here: tests/ and bench.py always load THIS file.  The tools that time -DOI_ABLATION builds set
# LIB_PATH explicitly before first use -- tools/_ablation.py.)

OI_HOST, OI_DEVICE = 0, 1
OI_MAX_DEPTH = 1024
OI_MAX_DIM = 1024
OI_BM25_BLOCK_DOCS = 32768
OI_N_CATALYST_KEYWORDS = 16
OI_COSINE_EXACT, OI_COSINE_SPLIT, OI_COSINE_SCREEN, OI_COSINE_SCREEN_COPY, OI_COSINE_SCREEN_STREAM = 0, 1, 2, 3, 4
OI_SCREEN_COPY_AUTO, OI_SCREEN_COPY_NEVER, OI_SCREEN_COPY_ALWAYS = 0, 1, 2

OI_ERR_INVALID_ARG = -1
OI_ERR_HIP = -2
OI_ERR_ANALYZER_MISMATCH = -3
OI_ERR_STATE = -5
OI_ERR_NO_DEVICE = -6
OI_ERR_UNSUPPORTED = -7
OI_ERR_OVERFLOW = -8
OI_ERR_COMM = -9
OI_COMM_ID_BYTES = 128


class OiError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("libopenintel_hip error %d: %s" % (code, message))
        self.code = code
        self.message = message


class SocialCounters(C.Structure):
    _fields_ = [
        ("total", C.c_uint64),
        ("by_source", C.c_uint64 * 2),
        ("bullish", C.c_uint64),
        ("bearish", C.c_uint64),
        ("neutral", C.c_uint64),
        ("spec_count", C.c_uint64),
        ("polarity_sum", C.c_double),
    ]


_P = C.c_void_p
_U32, _U64, _I = C.c_uint32, C.c_uint64, C.c_int

# name -> (restype, argtypes); every symbol include/openintel_hip.h declares
SIGNATURES = {
    "oi_abi_version": (_I, []),
    "oi_last_error": (C.c_char_p, []),
    "oi_create": (_I, [_I, C.POINTER(_P)]),
    "oi_create_like": (_I, [_P, C.POINTER(_P)]),
    "oi_destroy": (None, [_P]),
    "oi_set_stream": (_I, [_P, _P]),
    "oi_synchronize": (_I, [_P]),
    "oi_lexicon_analyze": (_I, [_P, _P, _P, _U64, _P, _P]),
    "oi_lexicon_analyze_device": (_I, [_P, _P, _P, _U64, _U64, _P, _P]),
    "oi_lexicon_summary_device": (_I, [_P, _P, _P, _U64, _U64, _P, C.c_double, _P, _P, C.POINTER(SocialCounters)]),
    "oi_workspace_bytes": (_I, [_P, C.POINTER(_U64), C.POINTER(_U64)]),
    "oi_set_overlap": (_I, [_P, _I]),
    "oi_set_screen_speculation": (_I, [_P, _I]),
    "oi_set_graph_replay": (_I, [_P, _I]),
    "oi_set_cosine_mode": (_I, [_P, _I]),
    "oi_catalyst_keyword": (C.c_char_p, [_U32]),
    "oi_headline_scan": (_I, [_P, _P, _P, _U64, _P, _U64, _P, _P, _U32, _P, _P, _P]),
    "oi_headline_scan_device": (_I, [_P, _P, _P, _U64, _U64, _P, _U64, _P, _P, _U32, _P, _P, _P]),
    "oi_headline_scan_rows": (_I, [_P, _P, _P, _U64, _P, _U32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "oi_social_summary": (_I, [_P, _P, _U64, _P, _P, _U64, C.c_double, _I, C.POINTER(SocialCounters)]),
    "oi_social_summary_segmented": (_I, [_P, _P, _P, _P, _U64, _P, _U64, C.c_double, _I, _P]),
    "oi_lexicon_scan_segments_device": (_I, [_P, _P, _P, _U64, _U64, _P, _P, _U64, C.c_double, _P, _P, _P]),
    "oi_index_create": (_I, [_P, _U64, _U32, _U32, _U32, C.POINTER(_P)]),
    "oi_index_destroy": (None, [_P]),
    "oi_index_view": (_I, [_P, _P, C.POINTER(_P)]),
    "oi_index_set_embeddings": (_I, [_P, _P, _I, _I]),
    "oi_index_set_embeddings_bf16": (_I, [_P, _P, _I]),
    "oi_index_set_forward": (_I, [_P, _P, _P, _I]),
    "oi_index_local_stats": (_I, [_P, C.POINTER(_U64), _P]),
    "oi_index_set_max_query_terms": (_I, [_P, _U32]),
    "oi_index_set_bm25_mode": (_I, [_P, _I]),
    "oi_index_long_rows": (_I, [_P, C.POINTER(C.c_uint32)]),
    "oi_index_set_screen_copy": (_I, [_P, _I]),
    "oi_index_bytes": (_I, [_P, C.POINTER(_U64), C.POINTER(_U64), C.POINTER(_U64)]),
    "oi_index_finalize": (_I, [_P, _U64, _U64, _P]),
    "oi_search_lists": (_I, [_P, _P, _P, _P, _U32, _U32, _I, _P, _P, _P, _P, _P, _P]),
    "oi_search_lists_packed": (_I, [_P, _P, _P, _P, _U32, _U32, _I, _P]),
    "oi_fuse_packed": (_I, [_P, _P, _U32, _U32, _U32, _U32, _I, _P, _P, _P]),
    "oi_merge_lists": (_I, [_P, _P, _P, _P, _U32, _U32, _U32, _I, _P, _P, _P]),
    "oi_rrf_fuse": (_I, [_P, _P, _P, _P, _P, _U32, _U32, _U32, _I, _P, _P, _P]),
    "oi_search": (_I, [_P, _P, _P, _P, _U32, _U32, _U32, _I, _P, _P, _P]),
    "oi_comm_unique_id": (_I, [_P]),
    "oi_comm_create": (_I, [_P, _P, _U32, _U32, C.POINTER(_P)]),
    "oi_comm_destroy": (None, [_P]),
    "oi_index_finalize_sharded": (_I, [_P, _P]),
    "oi_search_sharded": (_I, [_P, _P, _P, _P, _P, _U32, _U32, _U32, _I, _P, _P, _P]),
    "oi_pipeline_create": (_I, [_P, _P, _U32, _U32, _U32, _U32, _U32, C.POINTER(_P)]),
    "oi_pipeline_destroy": (None, [_P]),
    "oi_pipeline_submit": (_I, [_P, _P, _P, _P, _U32, _I, _P, _P, _P, C.POINTER(_U64)]),
    "oi_pipeline_wait": (_I, [_P, _U64, _I]),
    "oi_pipeline_drain": (_I, [_P]),
    "oi_pipeline_workspace_bytes": (_I, [_P, C.POINTER(_U64), C.POINTER(_U64)]),
    "oi_pipeline_concurrent_streams": (_I, [_P, C.POINTER(_U32), C.POINTER(_U32)]),
    "oi_pipeline_profile_reset": (_I, [_P, _I]),
    "oi_pipeline_profile_read": (_I, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_U64)]),
    "oi_screen_probe": (_I, [_P, _P, _U32, _U64, _U32, _P, _P]),
    "oi_profile_reset": (_I, [_P, _I]),
    "oi_profile_read": (_I, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_U64)]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library and bind every declared symbol.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libopenintel_hip.so is not built (%s). Run `python -m openintel_amd.build`. "
            "openintel_amd has no CPU fallback." % LIB_PATH)
    # torch bundles its own HIP runtime (same soname, libamdhip64.so.7).  Two HIP runtimes in one
    # process cannot share a GPU, so when torch is present it must be loaded FIRST: the dynamic
    # linker then resolves this library's libamdhip64.so.7 to the copy torch already mapped, and
    # torch tensors / streams are valid arguments.  A host without torch (the Rust shim of
    # INTEGRATION.md) simply gets the system ROCm runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().oi_last_error()
        raise OiError(rc, msg.decode("utf-8", "replace") if msg else "")


def ptr(x):
    """Device or host address of a torch tensor / numpy array / None."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(x.ctypes.data)
