"""fp16 mode of the reference (mmdet/core/fp16): `wrap_fp16_model` (hooks.py:86-105), `patch_norm_fp32` (:108-127),
`auto_fp16` / `force_fp32` (decorators.py:8-63, :66-123) and `cast_tensor_type` (utils.py:7-23), same names and
argument meaning.  Parameters and activations are stored as half exactly where the reference stores them; the HIP
kernels compute in fp32 (more accurate than the reference's half kernels, never less), and every half<->float
conversion of a GPU tensor goes through libarfe_hip (arfe_cast_*).  Loss scaling / the fp32 master copy of
Fp16OptimizerHook belong to the training harness and are not on the path."""
import functools
import inspect
from collections import abc

import numpy as np
import torch
import torch.nn as nn

from . import functional as Fn


def _to(t, dst_type):
    if t.is_cuda and {t.dtype, dst_type} == {torch.float16, torch.float32}:
        return Fn.to_f16(t) if dst_type == torch.float16 else Fn.to_f32(t)
    return t.to(dst_type)


def cast_tensor_type(inputs, src_type, dst_type):
    """Converts every tensor inside a (nested) container to `dst_type`.  `src_type` is accepted for signature
    compatibility only: like the reference's helper, the conversion does not look at the dtype a tensor has."""
    if torch.is_tensor(inputs):
        return _to(inputs, dst_type)
    if isinstance(inputs, (str, np.ndarray)) or not isinstance(inputs, abc.Iterable):
        return inputs
    if isinstance(inputs, abc.Mapping):
        return type(inputs)({key: cast_tensor_type(val, src_type, dst_type) for key, val in inputs.items()})
    return type(inputs)(cast_tensor_type(val, src_type, dst_type) for val in inputs)


def _cast_call(old_func, apply_to, src, dst, out_cast, args, kwargs):
    names = inspect.getfullargspec(old_func).args
    wanted = names if apply_to is None else apply_to
    new_args = [cast_tensor_type(a, src, dst) if i < len(names) and names[i] in wanted else a
                for i, a in enumerate(args)]
    new_kwargs = {k: cast_tensor_type(v, src, dst) if k in wanted else v for k, v in kwargs.items()}
    output = old_func(*new_args, **new_kwargs)
    if out_cast:
        output = cast_tensor_type(output, dst, src)
    return output


def _decorator(kind, apply_to, src, dst, out_cast):
    def wrapper(old_func):
        @functools.wraps(old_func)
        def new_func(*args, **kwargs):
            if not isinstance(args[0], nn.Module):
                raise TypeError(f'@{kind} can only be used to decorate the method of nn.Module')
            if not (hasattr(args[0], 'fp16_enabled') and args[0].fp16_enabled):
                return old_func(*args, **kwargs)
            return _cast_call(old_func, apply_to, src, dst, out_cast, args, kwargs)
        return new_func
    return wrapper


def auto_fp16(apply_to=None, out_fp32=False):
    """With `self.fp16_enabled`, float tensors among the named arguments become half (out_fp32: half outputs -> float)."""
    return _decorator('auto_fp16', apply_to, torch.float, torch.half, out_fp32)


def force_fp32(apply_to=None, out_fp16=False):
    """With `self.fp16_enabled`, half tensors among the named arguments become float (out_fp16: float outputs -> half)."""
    return _decorator('force_fp32', apply_to, torch.half, torch.float, out_fp16)


def patch_forward_method(func, src_type, dst_type, convert_output=True):
    """forward wrapper: arguments converted to dst_type, the result (optionally) back to src_type."""
    def new_forward(*args, **kwargs):
        out = func(*cast_tensor_type(args, src_type, dst_type), **cast_tensor_type(kwargs, src_type, dst_type))
        return cast_tensor_type(out, dst_type, src_type) if convert_output else out
    return new_forward


def patch_norm_fp32(module):
    """Normalisation layers stay in fp32 inside a half model; GroupNorm additionally gets a half<->float forward."""
    for m in module.modules():
        if isinstance(m, (nn.modules.batchnorm._BatchNorm, nn.GroupNorm)):
            m.float()
            if isinstance(m, nn.GroupNorm):
                m.forward = patch_forward_method(m.forward, torch.half, torch.float)
    return module


def wrap_fp16_model(model):
    """Half parameters (norm layers excepted) and `fp16_enabled = True` on every module that declares the attribute."""
    patch_norm_fp32(model.half())
    for m in model.modules():
        if hasattr(m, 'fp16_enabled'):
            m.fp16_enabled = True
