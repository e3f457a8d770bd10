"""ctypes binding of libarfe_hip.so (C ABI declared in include/arfe_hip.h).

The HIP library IS the product path: there is no CPU or PyTorch fallback.  If the shared library is
missing (or was built for another ABI version) importing this module's `lib()` raises immediately.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARFE_LIB", os.path.join(_HERE, "libarfe_hip.so"))   # ARFE_LIB: development builds only
MAX_LEVELS = 8
ABI_VERSION = 3

c_f32p = ctypes.c_void_p
_i32, _i64, _vp, _f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_float


class Pyramid(ctypes.Structure):
    _fields_ = [("num_levels", _i32), ("refine_level", _i32), ("batch", _i32), ("channels", _i32),
                ("height", _i32 * MAX_LEVELS), ("width", _i32 * MAX_LEVELS), ("storage", _i32)]

    @staticmethod
    def of(tensors, refine_level=0):
        p = Pyramid()
        assert 1 <= len(tensors) <= MAX_LEVELS
        p.num_levels = len(tensors)
        p.refine_level = refine_level
        p.batch, p.channels = tensors[0].shape[0], tensors[0].shape[1]
        p.storage = 1 if tensors[0].dtype == torch.float16 else 0      # what the per-level activation arrays hold
        for i, t in enumerate(tensors):
            assert t.shape[0] == p.batch and t.shape[1] == p.channels, "levels disagree on batch/channels"
            p.height[i], p.width[i] = t.shape[2], t.shape[3]
        return p


_PP = ctypes.POINTER(Pyramid)
# name -> (restype, argtypes); mirrors include/arfe_hip.h one to one
SIGNATURES = {
    "arfe_last_error": (ctypes.c_char_p, []),
    "arfe_abi_version": (_i32, []),
    "arfe_set_operand_precision": (_i32, [_i32]),
    "arfe_get_operand_precision": (_i32, []),
    "arfe_fpn_half_storage_supported": (_i32, [_PP]),
    "arfe_fpn_gather_fwd": (_i32, [_PP, _vp, _vp, _vp, _vp]),
    "arfe_fpn_gatemap_workspace_bytes": (_i64, [_PP]),
    "arfe_fpn_gatemap_fwd": (_i32, [_PP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "arfe_fpn_gather_gate_workspace_bytes": (_i64, [_PP]),
    "arfe_fpn_gather_gate_fwd": (_i32, [_PP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "arfe_fpn_scatter_fwd": (_i32, [_PP, _vp, _vp, _vp, _vp, _vp, _vp]),
    "arfe_fpn_scatter_bwd": (_i32, [_PP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "arfe_fpn_input_bwd": (_i32, [_PP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "arfe_fpn_gate_wgrad": (_i32, [_PP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "arfe_nonlocal_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "arfe_nonlocal_fwd": (_i32, [_i32, _i32, _i32] + [_vp] * 15),
    "arfe_nonlocal_bwd": (_i32, [_i32, _i32, _i32] + [_vp] * 21),
    "arfe_nonlocal_bwd_wout": (_i32, [_i32, _i32, _i32] + [_vp] * 5),
    "arfe_nonlocal_bwd_data": (_i32, [_i32, _i32, _i32] + [_vp] * 12),
    "arfe_nonlocal_bwd_dtpg": (_i32, [_i32, _i32, _i32] + [_vp] * 12),
    "arfe_nonlocal_bwd_dm": (_i32, [_i32, _i32, _i32] + [_vp] * 12),
    "arfe_nonlocal_bwd_wcat": (_i32, [_i32, _i32, _i32] + [_vp] * 9),
    "arfe_rff_make_rois": (_i32, [_i32, _vp, _vp, _vp, _vp]),
    "arfe_map_roi_levels": (_i32, [_i32, _vp, _i32, _f32, _vp, _vp]),
    "arfe_rff_roi_sets": (_i32, [_i32, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "arfe_roialign_fwd_workspace_bytes": (_i64, [_i32, _i32]),
    "arfe_roialign_fwd": (_i32, [_PP, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp]),
    "arfe_roialign_fwd_prepare": (_i32, [_PP, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    "arfe_roialign_fwd_group": (_i32, [_PP, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _i32, _i32, _vp]),
    "arfe_roialign_fwd_h": (_i32, [_PP, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp]),
    "arfe_roialign_bwd_prepared_h": (_i32, [_PP, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp]),
    "arfe_roialign_bwd_workspace_bytes": (_i64, [_PP, _i32, _i32]),
    "arfe_roialign_bwd": (_i32, [_PP, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp]),
    "arfe_roialign_bwd_prepare": (_i32, [_PP, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    "arfe_roialign_bwd_prepared": (_i32, [_PP, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp]),
    "arfe_roialign_bwd_prezero": (_i32, [_PP, _vp, _vp, _i32, _i32, _vp, _vp]),
    "arfe_roialign_bwd_apply": (_i32, [_PP, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp]),
    "arfe_conv3x3_packed_bytes": (_i64, [_i32]),
    "arfe_conv3x3_workspace_bytes": (_i64, [_i32]),
    "arfe_conv3x3_pack_weights": (_i32, [_i32, _vp, _vp, _vp]),
    "arfe_conv3x3_fwd": (_i32, [_i32, _i32, _vp, _i64, _vp, _vp, _i32, _vp, _i64, _vp]),
    "arfe_conv3x3_fwd_fuse": (_i32, [_i32, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "arfe_conv3x3_bwd_data": (_i32, [_i32, _i32, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "arfe_conv3x3_bwd_weight": (_i32, [_i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "arfe_rff_fuse_fwd": (_i32, [_i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp]),
    "arfe_rff_fuse_bwd": (_i32, [_i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "arfe_linear_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "arfe_linear_fwd": (_i32, [_i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _vp]),
    "arfe_linear_bwd": (_i32, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    "arfe_conv1x1_fwd": (_i32, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "arfe_conv1x1_bwd_data": (_i32, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "arfe_conv1x1_bwd_weight": (_i32, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "arfe_convmap3x3_fwd": (_i32, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "arfe_convmap3x3_bwd_data": (_i32, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "arfe_convmap3x3_bwd_weight": (_i32, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "arfe_nearest_add_fwd": (_i32, [_i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "arfe_nearest_add_bwd": (_i32, [_i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "arfe_subsample2": (_i32, [_i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "arfe_cast_multi": (_i32, [_i32, _vp, _vp, _vp, _i32, _vp]),
    "arfe_cast_f16_to_f32": (_i32, [_i64, _vp, _vp, _vp]),
    "arfe_cast_f32_to_f16": (_i32, [_i64, _vp, _vp, _vp]),
}

_lib = None


class ArfeHipError(RuntimeError):
    pass


def lib():
    """The loaded C-ABI library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ArfeHipError(
                f"{LIB_PATH} is missing: build it with `python -m arfe_amd.build` (hipcc, gfx950). "
                "arfe_amd has no CPU/PyTorch fallback for the AR-FPN / AR-RFF path.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        v = l.arfe_abi_version()
        if v != ABI_VERSION:
            raise ArfeHipError(f"libarfe_hip ABI {v} != binding ABI {ABI_VERSION}; rebuild the library")
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().arfe_last_error()
        raise ArfeHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t):
    return 0 if t is None else t.data_ptr()


def ptr_array(tensors):
    """A host array of device pointers (read by the library at launch time)."""
    arr = (ctypes.c_void_p * MAX_LEVELS)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(dtype, *tensors):
    for t in tensors:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
            raise ArfeHipError(f"expected a contiguous {dtype} tensor on the GPU, got {t.dtype} {t.device} "
                               f"contiguous={t.is_contiguous()}")


def require_f32_cuda(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ArfeHipError(f"expected a contiguous fp32 tensor on the GPU, got {t.dtype} {t.device} "
                               f"contiguous={t.is_contiguous()}")
