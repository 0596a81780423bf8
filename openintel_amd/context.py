"""HipContext -- owns an oi_ctx (device, stream, workspaces) of libopenintel_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Optional

from . import _lib


class HipContext:
    """One per process/GPU.  Thread-safe: the library serialises calls on a ctx."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.oi_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)
        if stream is not None:
            self.set_stream(stream)

    @classmethod
    def like(cls, other: "HipContext") -> "HipContext":
        """A new context on `other`'s device with its settings (cosine mode, overlap, graph replay) and nothing else of it:
        its own stream and workspaces (oi_create_like).  What a pipeline lane or a calibration trial wants."""
        self = cls.__new__(cls)
        self.lib = other.lib
        h = C.c_void_p()
        _lib.check(self.lib.oi_create_like(other.handle, C.byref(h)))
        self.handle = h
        self.device = other.device
        return self

    def set_stream(self, stream) -> None:
        """`stream`: a hipStream_t as int, or a torch.cuda.Stream (its .cuda_stream is used)."""
        raw = getattr(stream, "cuda_stream", stream)
        _lib.check(self.lib.oi_set_stream(self.handle, C.c_void_p(int(raw) if raw else None)))

    def use_torch_current_stream(self) -> None:
        import torch
        self.set_stream(torch.cuda.current_stream(self.device))

    def synchronize(self) -> None:
        _lib.check(self.lib.oi_synchronize(self.handle))

    def set_cosine_mode(self, mode: int) -> None:
        """_lib.OI_COSINE_SCREEN (default: bf16 screen with a proven bound + exact f32 rescoring, HBM-bound; the screen
        streams the index's bf16 screening copy when it holds one, the f32 rows otherwise), _lib.OI_COSINE_EXACT (f32 MFMA
        for every row), _lib.OI_COSINE_SPLIT (six bf16 MFMAs per f32 product), _lib.OI_COSINE_SCREEN_COPY (screen; a missing
        copy is made on first use) or _lib.OI_COSINE_SCREEN_STREAM (screen, always over the f32 rows)."""
        _lib.check(self.lib.oi_set_cosine_mode(self.handle, int(mode)))

    def set_overlap(self, enable: bool) -> None:
        """BM25 leg of a hybrid query beside the cosine leg on a side stream (default) or after it."""
        _lib.check(self.lib.oi_set_overlap(self.handle, 1 if enable else 0))

    def set_screen_speculation(self, enable: bool) -> None:
        """Speculative screen thresholds (oi_set_screen_speculation; default on): predicted from the rows seen so far, checked at
        the end, the exact pipeline behind a failed check.  The lists do not depend on it."""
        _lib.check(self.lib.oi_set_screen_speculation(self.handle, 1 if enable else 0))

    def speculation_state(self):
        """(failed checks seen, searches that speculated) since the ctx was created."""
        f, n = self.profile_read("spec_state")
        return int(f), int(n)

    def set_graph_replay(self, enable: bool) -> None:
        """Capture repeated device-buffer query calls into hipGraphs and replay them with one launch (oi_set_graph_replay):
        same kernels, same results, ~0.3 ms less host time per call.  The caller keeps using the same buffers."""
        _lib.check(self.lib.oi_set_graph_replay(self.handle, 1 if enable else 0))

    def graph_stats(self):
        """(replays, captures) since the ctx was created."""
        return self.profile_read("graph_replays")[1], self.profile_read("graph_captures")[1]

    def workspace_bytes(self):
        """(HBM bytes of this context's workspaces, page-locked host bytes of its staging buffers) right now."""
        d, h = C.c_uint64(), C.c_uint64()
        _lib.check(self.lib.oi_workspace_bytes(self.handle, C.byref(d), C.byref(h)))
        return int(d.value), int(h.value)

    # ---- HIP-event kernel timing (bench.py)
    def profile_reset(self, enable=True) -> None:
        """enable: False/0 off, True/1 every tagged launch, 2 only the cosine scorer's launches."""
        _lib.check(self.lib.oi_profile_reset(self.handle, int(enable)))

    def profile_read(self, tag: str):
        ms, n = C.c_double(), C.c_uint64()
        _lib.check(self.lib.oi_profile_read(self.handle, tag.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.oi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
