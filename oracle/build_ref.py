"""TEST INFRASTRUCTURE. Build recipe for the parts of the oracle that are compiled:

  oracle/_build/liboracle.so          <- oracle/roi_align_oracle.c (our C restatement; gcc)
  oracle/_ref/arfe_ref_roi_align*.so  <- the REFERENCE's own RoIAlign CPU sources, compiled where
                                         they lie under /root/reference (never copied):
                                           mmdet/ops/roi_align/src/roi_align_ext.cpp
                                           mmdet/ops/roi_align/src/cpu/roi_align_v2.cpp
                                         (torch C++ extension, no WITH_CUDA; g++ via torch's own
                                         cpp_extension driver — the reference's build system is not run)

`oracle/_ref/` is git-ignored (outputs only) but travels to the GPU box with the snapshot.
The second target is skipped when /root/reference is absent (GPU box: prebuilt file is used).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("ARFE_REFERENCE_ROOT", "/root/reference")


def build_c_oracle(force=False):
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(HERE, "roi_align_oracle.c")
    out = os.path.join(out_dir, "liboracle.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-o", out, src, "-lm"])
    return out


def build_ref_roi_align(force=False):
    srcs = [os.path.join(REF_ROOT, "mmdet/ops/roi_align/src/roi_align_ext.cpp"),
            os.path.join(REF_ROOT, "mmdet/ops/roi_align/src/cpu/roi_align_v2.cpp")]
    if not all(os.path.exists(s) for s in srcs):
        return None
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    have = [f for f in os.listdir(out_dir) if f.startswith("arfe_ref_roi_align") and f.endswith(".so")]
    if have and not force:
        return os.path.join(out_dir, have[0])
    from torch.utils.cpp_extension import load
    load(name="arfe_ref_roi_align", sources=srcs, build_directory=out_dir, verbose=False,
         extra_cflags=["-O2"], is_python_module=True)
    have = [f for f in os.listdir(out_dir) if f.startswith("arfe_ref_roi_align") and f.endswith(".so")]
    return os.path.join(out_dir, have[0]) if have else None


if __name__ == "__main__":
    force = "--force" in sys.argv
    print("c oracle :", build_c_oracle(force))
    print("reference:", build_ref_roi_align(force))
