"""NonLocal2D with the interface of the reference's mmdet/ops/non_local.py:6-105, computing through
libarfe_hip (arfe_nonlocal_fwd / arfe_nonlocal_bwd)."""
import torch
import torch.nn as nn

from . import functional as Fn
from .cnn import ConvModule, constant_init, normal_init


class _NonLocalFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, wt, bt, wp, bp, wg, bg, wo, bo, half_ops=False):
        m = m.contiguous()
        ctx.half_ops = bool(half_ops)
        with Fn.lib_precision(ctx.half_ops):
            bsf, saved = Fn.nonlocal_fwd_op(m, wt, bt, wp, bp, wg, bg, wo, bo)
        ctx.save_for_backward(m, wt, wp, wg, wo, *saved)
        return bsf

    @staticmethod
    def backward(ctx, dbsf):
        m, wt, wp, wg, wo, tpg, P, y = ctx.saved_tensors
        with Fn.lib_precision(ctx.half_ops):
            dm, (dwt, dbt, dwp, dbp, dwg, dbg, dwo, dbo) = Fn.nonlocal_bwd_op(dbsf.contiguous(), m, wt, wp, wg, wo, (tpg, P, y))
        return dm, dwt, dbt, dwp, dbp, dwg, dbg, dwo, dbo, None


class NonLocal2D(nn.Module):
    def __init__(self, in_channels, reduction=2, use_scale=True, conv_cfg=None, norm_cfg=None,
                 mode='embedded_gaussian'):
        super().__init__()
        assert mode in ['embedded_gaussian', 'dot_product']
        if reduction != 1 or use_scale or mode != 'embedded_gaussian':
            raise NotImplementedError(
                "the AR-FPN path uses NonLocal2D(reduction=1, use_scale=False, mode='embedded_gaussian') "
                "(wfpn_dual_spatial.py:78-83); other variants have no HIP kernel")
        self.in_channels = in_channels
        self.reduction = reduction
        self.use_scale = use_scale
        self.inter_channels = in_channels // reduction
        self.mode = mode
        self.g = ConvModule(in_channels, self.inter_channels, kernel_size=1, act_cfg=None)
        self.theta = ConvModule(in_channels, self.inter_channels, kernel_size=1, act_cfg=None)
        self.phi = ConvModule(in_channels, self.inter_channels, kernel_size=1, act_cfg=None)
        self.conv_out = ConvModule(self.inter_channels, in_channels, kernel_size=1, conv_cfg=conv_cfg,
                                   norm_cfg=norm_cfg, act_cfg=None)
        self.init_weights()

    def init_weights(self, std=0.01, zeros_init=True):
        for m in [self.g, self.theta, self.phi]:
            normal_init(m.conv, std=std)
        if zeros_init:
            constant_init(self.conv_out.conv, 0)
        else:
            normal_init(self.conv_out.conv, std=std)

    def hip_params(self):
        c = self.in_channels
        return (self.theta.conv.weight.view(c, c), self.theta.conv.bias, self.phi.conv.weight.view(c, c),
                self.phi.conv.bias, self.g.conv.weight.view(c, c), self.g.conv.bias,
                self.conv_out.conv.weight.view(c, c), self.conv_out.conv.bias)

    def forward(self, x):
        (x,), p, finish = Fn.f32_boundary([x], self.hip_params(), 'NonLocal2D')
        return finish(_NonLocalFunction.apply(x, *p, finish.half_ops))
