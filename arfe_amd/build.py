"""Builds libarfe_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m arfe_amd.build            # incremental
    python -m arfe_amd.build --force

Each translation unit is compiled to an object (the two GEMM-template units take a minute each), in parallel, into
arfe_amd/csrc/_obj/ (git-ignored *.o), and only the units whose sources or shared headers changed are recompiled.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libarfe_hip.so")
SOURCES = ["api.hip", "fpn.hip", "fpn_stream.hip", "roi.hip", "gemm_ops.hip", "fpn_base.hip"]
HEADERS = ["common.hpp", "gemm_f32.hpp", "gemm_bf16x3.hpp", "conv_stage_once.hpp", "conv_wgrad_stage_once.hpp", os.path.join("..", "..", "include", "arfe_hip.h")]   # (fpn_stream.hip is in SOURCES: it is hashed with them)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"]


def sources_sha():
    """sha256 over the kernel sources and headers: profiles record it, bench.py quotes PMC traffic only when it matches."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _mtime(path):
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _header_time():
    return max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)


def _stale_objects():
    ht = _header_time()
    out = []
    for s in SOURCES:
        o = os.path.join(OBJ, s + ".o")
        if _mtime(o) < max(_mtime(os.path.join(CSRC, s)), ht):
            out.append(s)
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, verbose):
    cmd = [HIPCC] + CFLAGS + ["-c", os.path.join(CSRC, src), "-o", os.path.join(OBJ, src + ".o")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    todo = list(SOURCES) if force else _stale_objects()
    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), max(1, (os.cpu_count() or 2) // 2))) as ex:
            list(ex.map(lambda s: _compile(s, verbose), todo))
    cmd = [HIPCC] + LDFLAGS + ["-o", LIB] + [os.path.join(OBJ, s + ".o") for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
