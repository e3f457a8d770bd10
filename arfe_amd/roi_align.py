"""RoIAlign layer with the interface of the reference's mmdet/ops/roi_align/roi_align.py:79-154 (aligned=True
"v2" semantics only — the path's configuration).  Single feature map; the multi-level dispatch
lives in SingleRoIExtractor, which calls the fused multi-level kernel directly."""
import torch.nn as nn
from torch.nn.modules.utils import _pair

from . import functional as Fn


class RoIAlign(nn.Module):
    def __init__(self, out_size, spatial_scale, sample_num=0, use_torchvision=False, aligned=True):
        super().__init__()
        self.out_size = _pair(out_size)
        self.spatial_scale = float(spatial_scale)
        self.aligned = aligned
        self.sample_num = int(sample_num)
        self.use_torchvision = use_torchvision
        if use_torchvision or not aligned:
            raise NotImplementedError("only aligned=True (RoIAlign v2) has a HIP kernel")
        if self.out_size[0] != self.out_size[1]:
            raise NotImplementedError("square output only")
        stride = 1.0 / self.spatial_scale
        if abs(stride - round(stride)) > 1e-6:
            raise NotImplementedError("spatial_scale must be 1/integer stride")
        self._stride = int(round(stride))

    def forward(self, features, rois):
        assert rois.dim() == 2 and rois.size(1) == 5
        # one level: every RoI maps to level 0
        return Fn.RoIExtractFunction.apply((self._stride,), 1e30, self.out_size[0], self.sample_num, 1, False, rois, features)

    def __repr__(self):
        return (f"{self.__class__.__name__}(out_size={self.out_size}, spatial_scale={self.spatial_scale}, "
                f"sample_num={self.sample_num}, use_torchvision={self.use_torchvision}, aligned={self.aligned})")
