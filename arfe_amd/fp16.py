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
    if isinstance(inputs, torch.Tensor):
        return _to(inputs, dst_type)      # like the reference, every tensor is converted; src_type is not consulted
    if isinstance(inputs, (str, np.ndarray)):
        return inputs
    if isinstance(inputs, abc.Mapping):
        return type(inputs)({k: cast_tensor_type(v, src_type, dst_type) for k, v in inputs.items()})
    if isinstance(inputs, abc.Iterable):
        return type(inputs)(cast_tensor_type(item, src_type, dst_type) for item in inputs)
    return inputs


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
    def new_forward(*args, **kwargs):
        output = func(*cast_tensor_type(args, src_type, dst_type), **cast_tensor_type(kwargs, src_type, dst_type))
        if convert_output:
            output = cast_tensor_type(output, dst_type, src_type)
        return output
    return new_forward


def patch_norm_fp32(module):
    if isinstance(module, (nn.modules.batchnorm._BatchNorm, nn.GroupNorm)):
        module.float()
        if isinstance(module, nn.GroupNorm):
            module.forward = patch_forward_method(module.forward, torch.half, torch.float)
    for child in module.children():
        patch_norm_fp32(child)
    return module


def wrap_fp16_model(model):
    model.half()
    patch_norm_fp32(model)
    for m in model.modules():
        if hasattr(m, 'fp16_enabled'):
            m.fp16_enabled = True
