"""ResNet-50/101/152 backbone of the detector harness in plain PyTorch-ROCm (MIOpen convolutions): the part of BASELINE
configs #2-#5 upstream of the hot path.  Structure, constructor arguments and state_dict keys (`conv1`, `bn1`,
`layer{1..4}.{i}.conv{1,2,3}`, `.bn{1,2,3}`, `.downsample.{0,1}`) follow the reference's
mmdet/models/backbones/resnet.py:118-347 (Bottleneck), :350-705 (ResNet) for the configuration the AR configs use:
style='pytorch' (stride on the 3x3), BN, no DCN / plugins / deep stem / avg_down."""
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from ..registry import BACKBONES


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch'):
        super().__init__()
        assert style in ('pytorch', 'caffe')
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=s1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=s2, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    @property
    def norm3(self):
        return self.bn3

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + (x if self.downsample is None else self.downsample(x)))


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth, in_channels=3, base_channels=64, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(0, 1, 2, 3), style='pytorch', deep_stem=False, avg_down=False, frozen_stages=-1,
                 conv_cfg=None, norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False, zero_init_residual=True):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet (the harness builds the Bottleneck depths 50/101/152)')
        if deep_stem or avg_down or conv_cfg is not None or dcn is not None or plugins is not None or with_cp:
            raise NotImplementedError("harness ResNet: plain conv/BN bottleneck stages only")
        if norm_cfg.get('type', 'BN') != 'BN':
            raise NotImplementedError(f"norm {norm_cfg}")
        assert 1 <= num_stages <= 4 and len(strides) == len(dilations) == num_stages and max(out_indices) < num_stages
        self.depth, self.out_indices, self.frozen_stages = depth, out_indices, frozen_stages
        self.norm_eval, self.zero_init_residual = norm_eval, zero_init_residual
        self.conv1 = nn.Conv2d(in_channels, base_channels, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(base_channels)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        bn_grad = norm_cfg.get('requires_grad', True)
        self.res_layers = []
        inplanes = base_channels
        for i, nblocks in enumerate(self.arch_settings[depth][:num_stages]):
            planes = base_channels * 2 ** i
            blocks = []
            for b in range(nblocks):
                stride = strides[i] if b == 0 else 1
                down = None
                if b == 0 and (stride != 1 or inplanes != planes * Bottleneck.expansion):
                    down = nn.Sequential(nn.Conv2d(inplanes, planes * Bottleneck.expansion, 1, stride=stride, bias=False),
                                         nn.BatchNorm2d(planes * Bottleneck.expansion))
                blocks.append(Bottleneck(inplanes, planes, stride, dilations[i], down, style))
                inplanes = planes * Bottleneck.expansion
            name = f'layer{i + 1}'
            self.add_module(name, nn.Sequential(*blocks))
            self.res_layers.append(name)
        if not bn_grad:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    for p in m.parameters():
                        p.requires_grad = False
        self.feat_dim = inplanes
        self._freeze_stages()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def init_weights(self, pretrained=None):
        if pretrained is not None:
            # the reference's configs name 'torchvision://resnet50'; there is no network and no checkpoint on these
            # machines, so the harness says so and initialises randomly (resnet.py:631-655: the `pretrained is None` branch)
            import warnings
            warnings.warn(f"ResNet: pretrained={pretrained!r} cannot be fetched here; random initialisation instead")
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, a=0, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, _BatchNorm):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
        return self
