"""Builds libopenintel_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m openintel_amd.build [--force] [--save-temps]

One translation unit per .hip file, objects cached under openintel_amd/csrc/_obj/, linked
into openintel_amd/libopenintel_hip.so (git-ignored; ships to the GPU box with the tree).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libopenintel_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off: BM25/RRF scores must round exactly like the written expression
# (bit-exact rank parity); the cosine kernels ask for FMA explicitly where they want it.
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-Wall", "-Wno-unused-function", "-Wno-unused-result",
    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
] + ([x for x in os.environ.get("OI_EXTRA_HIPCC_FLAGS", "").split() if x])


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "openintel_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool, extra):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    spath = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(spath)
            and os.path.getmtime(obj) >= _deps_mtime()):
        return obj
    cmd = [HIPCC, *FLAGS, *extra, "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr[-6000:]))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, save_temps: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    extra = ["-save-temps"] if save_temps else []
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, extra), srcs))
    if (force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True)
