"""Builds libarfe_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m arfe_amd.build            # incremental
    python -m arfe_amd.build --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libarfe_hip.so")
SOURCES = ["api.hip", "fpn.hip", "roi.hip", "gemm_ops.hip", "fpn_base.hip"]
HEADERS = ["common.hpp", "gemm_f32.hpp", "gemm_bf16x3.hpp", os.path.join("..", "..", "include", "arfe_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
         "-Wno-unused-result"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
