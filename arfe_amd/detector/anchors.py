"""Anchor grid of the detector harness (reference mmdet/core/anchor/anchor_generator.py:9-275, core/anchor/utils.py:4-36)."""
import math

import torch
from torch.nn.modules.utils import _pair

from ..registry import ANCHOR_GENERATORS


@ANCHOR_GENERATORS.register_module()
class AnchorGenerator:
    """Per level: `len(ratios) * len(scales)` base anchors of size base_size * scale, aspect h/w = ratio, centred at
    center_offset * base_size (0 in mmdet 2.0), shifted to every stride-spaced grid position, row-major, the
    base anchors of one position adjacent.

    >>> AnchorGenerator([16], [1.], [1.], [9]).grid_anchors([(2, 2)], device='cpu')[0].tolist()   # anchor_generator.py:41-48
    [[-4.5, -4.5, 4.5, 4.5], [11.5, -4.5, 20.5, 4.5], [-4.5, 11.5, 4.5, 20.5], [11.5, 11.5, 20.5, 20.5]]
    """

    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True, octave_base_scale=None,
                 scales_per_octave=None, centers=None, center_offset=0.):
        if center_offset != 0:
            assert centers is None, f'center cannot be set when center_offset!=0, {centers} is given.'
        if not (0 <= center_offset <= 1):
            raise ValueError(f'center_offset should be in range [0, 1], {center_offset} is given.')
        if centers is not None:
            assert len(centers) == len(strides)
        self.strides = [_pair(s) for s in strides]
        self.base_sizes = [min(s) for s in self.strides] if base_sizes is None else base_sizes
        assert len(self.base_sizes) == len(self.strides)
        octave = octave_base_scale is not None and scales_per_octave is not None
        assert octave ^ (scales is not None), \
            'scales and octave_base_scale with scales_per_octave cannot be set at the same time'
        if scales is not None:
            self.scales = torch.Tensor(scales)
        else:
            self.scales = torch.Tensor([2 ** (i / scales_per_octave) * octave_base_scale for i in range(scales_per_octave)])
        self.octave_base_scale, self.scales_per_octave = octave_base_scale, scales_per_octave
        self.ratios = torch.Tensor(ratios)
        self.scale_major, self.centers, self.center_offset = scale_major, centers, center_offset
        self.base_anchors = self.gen_base_anchors()

    @property
    def num_base_anchors(self):
        return [b.size(0) for b in self.base_anchors]

    @property
    def num_levels(self):
        return len(self.strides)

    def gen_base_anchors(self):
        return [self.gen_single_level_base_anchors(size, self.scales, self.ratios,
                                                   None if self.centers is None else self.centers[i])
                for i, size in enumerate(self.base_sizes)]

    def gen_single_level_base_anchors(self, base_size, scales, ratios, center=None):
        cx, cy = (self.center_offset * base_size, self.center_offset * base_size) if center is None else center
        h_ratios = torch.sqrt(ratios)
        w_ratios = 1 / h_ratios
        if self.scale_major:        # anchors of one ratio adjacent
            ws = (base_size * w_ratios[:, None] * scales[None, :]).view(-1)
            hs = (base_size * h_ratios[:, None] * scales[None, :]).view(-1)
        else:
            ws = (base_size * scales[:, None] * w_ratios[None, :]).view(-1)
            hs = (base_size * scales[:, None] * h_ratios[None, :]).view(-1)
        return torch.stack([cx - 0.5 * ws, cy - 0.5 * hs, cx + 0.5 * ws, cy + 0.5 * hs], dim=-1)

    def grid_anchors(self, featmap_sizes, device='cuda'):
        assert self.num_levels == len(featmap_sizes)
        return [self.single_level_grid_anchors(self.base_anchors[i].to(device), featmap_sizes[i], self.strides[i], device)
                for i in range(self.num_levels)]

    def single_level_grid_anchors(self, base_anchors, featmap_size, stride=(16, 16), device='cuda'):
        feat_h, feat_w = featmap_size
        sx = (torch.arange(0, feat_w, device=device) * stride[0]).repeat(feat_h)
        sy = (torch.arange(0, feat_h, device=device) * stride[1]).repeat_interleave(feat_w)
        shifts = torch.stack([sx, sy, sx, sy], dim=-1).type_as(base_anchors)
        return (base_anchors[None, :, :] + shifts[:, None, :]).view(-1, 4)

    def valid_flags(self, featmap_sizes, pad_shape, device='cuda'):
        """Anchors of grid positions inside the (padded) image (positions beyond ceil(pad / stride) are batch padding)."""
        assert self.num_levels == len(featmap_sizes)
        flags = []
        for i in range(self.num_levels):
            feat_h, feat_w = featmap_sizes[i]
            h, w = pad_shape[:2]
            vh = min(int(math.ceil(h / self.strides[i][0])), feat_h)
            vw = min(int(math.ceil(w / self.strides[i][1])), feat_w)
            flags.append(self.single_level_valid_flags((feat_h, feat_w), (vh, vw), self.num_base_anchors[i], device))
        return flags

    def single_level_valid_flags(self, featmap_size, valid_size, num_base_anchors, device='cuda'):
        feat_h, feat_w = featmap_size
        vh, vw = valid_size
        assert vh <= feat_h and vw <= feat_w
        vx = torch.arange(feat_w, device=device) < vw
        vy = torch.arange(feat_h, device=device) < vh
        valid = (vy[:, None] & vx[None, :]).view(-1)
        return valid[:, None].expand(valid.size(0), num_base_anchors).contiguous().view(-1)

    def __repr__(self):
        return (f'{self.__class__.__name__}(strides={self.strides}, ratios={self.ratios}, scales={self.scales}, '
                f'base_sizes={self.base_sizes}, scale_major={self.scale_major}, num_levels={self.num_levels}, '
                f'center_offset={self.center_offset})')


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    """core/anchor/utils.py:22-36: anchors that stay within `allowed_border` pixels of the image (any anchor if < 0)."""
    if allowed_border < 0:
        return valid_flags
    img_h, img_w = img_shape[:2]
    return (valid_flags & (flat_anchors[:, 0] >= -allowed_border) & (flat_anchors[:, 1] >= -allowed_border) &
            (flat_anchors[:, 2] < img_w + allowed_border) & (flat_anchors[:, 3] < img_h + allowed_border))


def images_to_levels(target, num_levels):
    """[per-image tensors over all anchors] -> [per-level tensors over the batch] (core/anchor/utils.py:4-19)."""
    target = torch.stack(target, 0)
    out, start = [], 0
    for n in num_levels:
        out.append(target[:, start:start + n])
        start += n
    return out


def unmap(data, count, inds, fill=0):
    """Values of the selected subset back at their positions among `count` items (core/utils/misc.py:28-38)."""
    ret = data.new_full((count, ) + tuple(data.size()[1:]), fill)
    ret[inds.type(torch.bool)] = data
    return ret
