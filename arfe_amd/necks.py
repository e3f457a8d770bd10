"""AR-FPN: `WFPNDualSpatial` with the constructor, registry name and state_dict keys of the reference
(mmdet/models/necks/wfpn_dual_spatial.py:10-137); forward/backward run in libarfe_hip."""
import torch.nn as nn

from . import functional as Fn
from .cnn import ConvModule, xavier_init
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
        return tuple(Fn.arfpn(list(inputs), w1, b1, w2, b2, *self.refine.hip_params(),
                              refine_level=self.refine_level))
