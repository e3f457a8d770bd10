"""AR-FPN: `WFPNDualSpatial` with the constructor, registry name and state_dict keys of the reference
(mmdet/models/necks/wfpn_dual_spatial.py:10-137); forward/backward run in libarfe_hip."""
import torch
import torch.nn as nn

from . import functional as Fn
from .cnn import ConvModule, xavier_init
from .fp16 import auto_fp16
from .non_local import NonLocal2D
from .registry import NECKS


@NECKS.register_module
class WFPNDualSpatial(nn.Module):
    def __init__(self, in_channels, num_levels, refine_level=2, conv_cfg=None, norm_cfg=None):
        super().__init__()
        self.in_channels = in_channels
        self.num_levels = num_levels
        self.conv_cfg = conv_cfg
        self.norm_cfg = norm_cfg
        self.refine_level = refine_level
        assert 0 <= refine_level < num_levels
        self.reduce_convs = nn.ModuleList()
        self.reduce_convs2 = nn.ModuleList()
        for _ in range(num_levels):
            self.reduce_convs.append(ConvModule(in_channels, 1, 3, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                                inplace=False))
            self.reduce_convs2.append(ConvModule(in_channels, 1, 3, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                                 inplace=False))
        self.refine = NonLocal2D(in_channels, reduction=1, use_scale=False, conv_cfg=conv_cfg, norm_cfg=norm_cfg)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')

    def forward(self, inputs):
        assert len(inputs) == self.num_levels
        w1 = [m.conv.weight for m in self.reduce_convs]
        b1 = [m.conv.bias for m in self.reduce_convs]
        w2 = [m.conv.weight for m in self.reduce_convs2]
        b2 = [m.conv.bias for m in self.reduce_convs2]
        # fp16 mode (wrap_fp16_model): half storage, fp32 arithmetic inside the library
        n = self.num_levels
        params = w1 + b1 + w2 + b2 + list(self.refine.hip_params())
        inputs = list(inputs)
        if (Fn.storage_dtype(inputs, params, 'WFPNDualSpatial') == torch.float16 and all(t.is_cuda for t in inputs)
                and Fn.fpn_half_storage_supported([t.contiguous() for t in inputs], self.refine_level)):
            # the streaming kernels read and write the half pyramid themselves (half the bytes, no conversion passes); only
            # the parameters are converted (one launch)
            p = Fn.to_f32_list(params)
            return tuple(Fn.arfpn(inputs, p[:n], p[n:2 * n], p[2 * n:3 * n], p[3 * n:4 * n], *p[4 * n:],
                                  refine_level=self.refine_level, half_ops=True))
        xs, p, finish = Fn.f32_boundary(inputs, params, 'WFPNDualSpatial')
        return finish(tuple(Fn.arfpn(xs, p[:n], p[n:2 * n], p[2 * n:3 * n], p[3 * n:4 * n], *p[4 * n:],
                                     refine_level=self.refine_level, half_ops=finish.half_ops)))


@NECKS.register_module()
class FPN(nn.Module):
    """Base FPN with the constructor, registry name and state_dict keys of the reference (mmdet/models/necks/fpn.py:10-164);
    SURVEY 8(f) row 2.  The configuration of the AR configs is implemented: no norm / activation in the ConvModules,
    nearest top-down upsampling, extra levels by `max_pool2d(1, stride=2)` (add_extra_convs=False)."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest')):
        super().__init__()
        assert isinstance(in_channels, list)
        if add_extra_convs:
            raise NotImplementedError("FPN extra levels by stride-2 convs (RetinaNet) are outside the AR configs")
        if act_cfg is not None or norm_cfg is not None or conv_cfg is not None:
            raise NotImplementedError("FPN ConvModules of the AR configs have no norm / activation")
        if dict(upsample_cfg) != dict(mode='nearest'):
            raise NotImplementedError(f"only upsample_cfg=dict(mode='nearest') has a kernel (got {upsample_cfg})")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_ins = len(in_channels)
        self.num_outs = num_outs
        self.relu_before_extra_convs = relu_before_extra_convs
        self.no_norm_on_lateral = no_norm_on_lateral
        self.fp16_enabled = False
        self.upsample_cfg = dict(upsample_cfg)
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= len(in_channels)
            assert num_outs == end_level - start_level
        self.start_level = start_level
        self.end_level = end_level
        self.add_extra_convs = add_extra_convs
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, act_cfg=None, inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, act_cfg=None, inplace=False))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')

    @auto_fp16()
    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        used = len(self.lateral_convs)
        params = [t for m in list(self.lateral_convs) + list(self.fpn_convs) for t in (m.conv.weight, m.conv.bias)]
        xs, p, finish = Fn.f32_boundary(inputs, params, 'FPN')
        h16 = finish.half_ops
        laterals = [Fn.conv1x1(xs[i + self.start_level], p[2 * i], p[2 * i + 1], h16) for i in range(used)]
        for i in range(used - 1, 0, -1):
            laterals[i - 1] = Fn.nearest_add(laterals[i - 1], laterals[i])
        outs = [Fn.convmap3x3(laterals[i], p[2 * (used + i)], p[2 * (used + i) + 1], h16) for i in range(used)]
        for _ in range(self.num_outs - used):
            outs.append(Fn.subsample2(outs[-1]))
        return finish(tuple(outs))
