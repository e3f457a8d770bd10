"""Parameter containers with the naming of mmcv.cnn.ConvModule (absent third-party; semantics restated in
SURVEY.md Appendix B.1): `<name>.conv.weight`, `<name>.conv.bias`, default activation ReLU, no norm on this path.
The nn.Conv2d inside is only a parameter holder — the arithmetic runs in the HIP kernels, never through it."""
import torch.nn as nn


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    assert distribution in ('uniform', 'normal')
    if distribution == 'uniform':
        nn.init.xavier_uniform_(module.weight, gain=gain)
    else:
        nn.init.xavier_normal_(module.weight, gain=gain)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    nn.init.normal_(module.weight, mean, std)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    nn.init.constant_(module.weight, val)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True, **kwargs):
        super().__init__()
        if conv_cfg is not None or norm_cfg is not None:
            raise NotImplementedError("arfe_amd implements the AR path's plain conv (conv_cfg=None, norm_cfg=None)")
        if stride != 1 or dilation != 1 or groups != 1:
            raise NotImplementedError("only stride 1 / dilation 1 / groups 1 convs are on the AR path")
        if act_cfg is not None and act_cfg.get('type') != 'ReLU':
            raise NotImplementedError(f"unsupported activation {act_cfg}")
        self.with_activation = act_cfg is not None
        self.with_bias = True if bias == 'auto' else bool(bias)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding, bias=self.with_bias)
        self.init_weights()

    def init_weights(self):
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)

    def forward(self, x):
        raise RuntimeError("ConvModule is a parameter container here; the owning module launches the HIP kernels")
