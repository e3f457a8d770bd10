"""TEST INFRASTRUCTURE (container-only). Loads the *unmodified* reference hot-path
files from /root/reference by path so golden vectors can be generated from them.

The reference (a vendored mmdetection 2.0.0 fork) cannot be imported as a package:
its third-party dependency `mmcv` (version unpinned, API of 0.5.x/0.6.x) is absent and
there is no network.  The few symbols the hot-path files need from mmcv / mmdet are
provided here as in-memory stand-ins whose semantics are restated from the call sites:

  * mmcv.cnn.ConvModule(cin, cout, k, padding=p, act_cfg=<default ReLU>|None, ...)
      == nn.Conv2d(bias=True) -> optional ReLU, parameters named `<name>.conv.weight/bias`
      (evidence: callers wanting no activation pass act_cfg=None explicitly,
      mmdet/ops/non_local.py:39-51).
  * mmcv.cnn.xavier_init / normal_init / constant_init: thin wrappers over torch.nn.init.
  * mmcv.utils.Registry / build_from_cfg: name -> class table.

Parity at the mmcv boundary is therefore "unpinned" (no reference test pins ConvModule);
everything downstream of that boundary is the reference's own code, executed as is.

Nothing in this file is shipped to the GPU box's product path; it is never imported by
`arfe_amd`.  /root/reference does not exist on the GPU box.
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("ARFE_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "mmdet"))


# ----------------------------------------------------------------------------- mmcv stand-in
class _ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, groups=1, bias='auto', conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), inplace=True, **kwargs):
        super().__init__()
        assert conv_cfg is None and norm_cfg is None, "hot path uses plain conv, no norm"
        self.with_activation = act_cfg is not None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                              padding=padding, dilation=dilation, groups=groups,
                              bias=True if bias == 'auto' else bias)
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        nn.init.constant_(self.conv.bias, 0)

    def forward(self, x):
        x = self.conv(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def _xavier_init(module, gain=1, bias=0, distribution='normal'):
    if distribution == 'uniform':
        nn.init.xavier_uniform_(module.weight, gain=gain)
    else:
        nn.init.xavier_normal_(module.weight, gain=gain)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def _normal_init(module, mean=0, std=1, bias=0):
    nn.init.normal_(module.weight, mean, std)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def _constant_init(module, val, bias=0):
    nn.init.constant_(module.weight, val)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


class _Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def get(self, key):
        return self.module_dict.get(key)

    def register_module(self, cls=None, **kw):
        if cls is None:
            return lambda c: self.register_module(c)
        self.module_dict[cls.__name__] = cls
        return cls


def _build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    return registry.get(args.pop('type'))(**args)


def _identity_decorator(*a, **k):
    def deco(f):
        return f
    return deco


_loaded = {}


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave as a package
    sys.modules[name] = m
    return m


def _install_stubs():
    if 'mmcv' in sys.modules and getattr(sys.modules['mmcv'], '_arfe_stub', False):
        return
    _mod('mmcv', _arfe_stub=True)
    _mod('mmcv.cnn', ConvModule=_ConvModule, xavier_init=_xavier_init,
         normal_init=_normal_init, constant_init=_constant_init)
    _mod('mmcv.utils', Registry=_Registry, build_from_cfg=_build_from_cfg)
    _mod('mmdet')
    _mod('mmdet.ops')
    _mod('mmdet.core', auto_fp16=_identity_decorator, force_fp32=_identity_decorator,
         build_bbox_coder=lambda cfg: None, multi_apply=None, multiclass_nms=None)
    _mod('mmdet.models')
    _mod('mmdet.models.losses', accuracy=None)
    _mod('mmdet.models.necks')
    _mod('mmdet.models.utils')
    _mod('mmdet.models.roi_heads')
    _mod('mmdet.models.roi_heads.bbox_heads')
    _mod('mmdet.models.roi_heads.roi_extractors')
    reg = {n: _Registry(n) for n in
           ('BACKBONES', 'NECKS', 'ROI_EXTRACTORS', 'SHARED_HEADS', 'HEADS', 'LOSSES', 'DETECTORS')}
    _mod('mmdet.models.builder', build_loss=lambda cfg: None, **reg)


def _load(modname, relpath):
    if modname in _loaded:
        return _loaded[modname]
    path = os.path.join(REF_ROOT, relpath)
    spec = importlib.util.spec_from_file_location(modname, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    _loaded[modname] = m
    return m


def load_reference():
    """Returns a namespace with the reference classes/functions of the hot path."""
    assert reference_available(), f"{REF_ROOT} not present (expected on the GPU box)"
    _install_stubs()
    # the reference's own fp16 decorators (pass-through unless a module's fp16_enabled is set); hooks.py, which holds
    # wrap_fp16_model, imports mmcv.runner and cannot be loaded - gen_golden restates its 5 lines
    _mod('mmdet.core.fp16')
    _load('mmdet.core.fp16.utils', 'mmdet/core/fp16/utils.py')
    deco = _load('mmdet.core.fp16.decorators', 'mmdet/core/fp16/decorators.py')
    sys.modules['mmdet.core'].auto_fp16 = deco.auto_fp16
    sys.modules['mmdet.core'].force_fp32 = deco.force_fp32
    nl = _load('mmdet.ops.non_local', 'mmdet/ops/non_local.py')
    sys.modules['mmdet.ops'].NonLocal2D = nl.NonLocal2D
    neck = _load('mmdet.models.necks.wfpn_dual_spatial', 'mmdet/models/necks/wfpn_dual_spatial.py')
    fpn = _load('mmdet.models.necks.fpn', 'mmdet/models/necks/fpn.py')
    add = _load('mmdet.models.utils.additional', 'mmdet/models/utils/additional.py')
    sys.modules['mmdet'].ops = sys.modules['mmdet.ops']
    bh = _load('mmdet.models.roi_heads.bbox_heads.bbox_head',
               'mmdet/models/roi_heads/bbox_heads/bbox_head.py')
    mh = _load('mmdet.models.roi_heads.bbox_heads.multirois_bbox_head',
               'mmdet/models/roi_heads/bbox_heads/multirois_bbox_head.py')
    ext = _load('mmdet.models.roi_heads.roi_extractors.single_level',
                'mmdet/models/roi_heads/roi_extractors/single_level.py')
    # the reference's python RoIAlign wrapper on top of the compiled reference sources (oracle/_ref)
    RoIAlign = None
    ext_mod = load_reference_roi_align()
    if ext_mod is not None:
        pkg = _mod('mmdet.ops.roi_align', roi_align_ext=ext_mod)
        ra = _load('mmdet.ops.roi_align.roi_align', 'mmdet/ops/roi_align/roi_align.py')
        pkg.RoIAlign = ra.RoIAlign
        sys.modules['mmdet.ops'].RoIAlign = ra.RoIAlign
        sys.modules['mmdet'].ops = sys.modules['mmdet.ops']
        RoIAlign = ra.RoIAlign
    ns = types.SimpleNamespace(
        RoIAlign=RoIAlign,
        roi_align_ext=ext_mod,
        NonLocal2D=nl.NonLocal2D,
        WFPNDualSpatial=neck.WFPNDualSpatial,
        FPN=fpn.FPN,
        get_adaptive_scale_rois=add.get_adaptive_scale_rois,
        BBoxHead=bh.BBoxHead,
        MultiBBoxHead=mh.MultiBBoxHead,
        MultiRoIsBBoxHead=mh.MultiRoIsBBoxHead,
        SingleRoIExtractor=ext.SingleRoIExtractor,
    )
    return ns


def load_reference_harness():
    """The reference's box / anchor / loss helpers of the detector harness, loaded by path like the
    hot-path files: core/bbox/{iou_calculators/iou2d_calculator, coder/delta_xywh_bbox_coder, assigners/max_iou_assigner},
    core/anchor/{anchor_generator, utils}, models/losses/{utils, cross_entropy_loss, smooth_l1_loss, accuracy}.
    Their `.builder` siblings (thin wrappers over the absent mmcv Registry) are stand-ins."""
    assert reference_available(), f"{REF_ROOT} not present"
    _install_stubs()
    sys.modules['mmcv'].is_tuple_of = lambda seq, t: isinstance(seq, tuple) and all(isinstance(s, t) for s in seq)
    regs = {n: _Registry(n) for n in ('IOU_CALCULATORS', 'BBOX_CODERS', 'BBOX_ASSIGNERS', 'BBOX_SAMPLERS', 'ANCHOR_GENERATORS')}
    _mod('mmdet.utils')
    um = _load('mmdet.utils.util_mixins', 'mmdet/utils/util_mixins.py')
    sys.modules['mmdet.utils'].util_mixins = um
    _mod('mmdet.core.bbox')
    _mod('mmdet.core.bbox.builder', BBOX_CODERS=regs['BBOX_CODERS'], BBOX_ASSIGNERS=regs['BBOX_ASSIGNERS'],
         BBOX_SAMPLERS=regs['BBOX_SAMPLERS'])
    ic = _mod('mmdet.core.bbox.iou_calculators')
    _mod('mmdet.core.bbox.iou_calculators.builder', IOU_CALCULATORS=regs['IOU_CALCULATORS'],
         build_iou_calculator=lambda cfg, default_args=None: _build_from_cfg(cfg, regs['IOU_CALCULATORS'], default_args))
    iou = _load('mmdet.core.bbox.iou_calculators.iou2d_calculator', 'mmdet/core/bbox/iou_calculators/iou2d_calculator.py')
    ic.build_iou_calculator = sys.modules['mmdet.core.bbox.iou_calculators.builder'].build_iou_calculator
    _mod('mmdet.core.bbox.coder')
    _load('mmdet.core.bbox.coder.base_bbox_coder', 'mmdet/core/bbox/coder/base_bbox_coder.py')
    coder = _load('mmdet.core.bbox.coder.delta_xywh_bbox_coder', 'mmdet/core/bbox/coder/delta_xywh_bbox_coder.py')
    _mod('mmdet.core.bbox.assigners')
    _load('mmdet.core.bbox.assigners.assign_result', 'mmdet/core/bbox/assigners/assign_result.py')
    _load('mmdet.core.bbox.assigners.base_assigner', 'mmdet/core/bbox/assigners/base_assigner.py')
    asg = _load('mmdet.core.bbox.assigners.max_iou_assigner', 'mmdet/core/bbox/assigners/max_iou_assigner.py')
    _mod('mmdet.core.anchor')
    _mod('mmdet.core.anchor.builder', ANCHOR_GENERATORS=regs['ANCHOR_GENERATORS'])
    anc = _load('mmdet.core.anchor.anchor_generator', 'mmdet/core/anchor/anchor_generator.py')
    autil = _load('mmdet.core.anchor.utils', 'mmdet/core/anchor/utils.py')
    _load('mmdet.models.losses.utils', 'mmdet/models/losses/utils.py')
    ce = _load('mmdet.models.losses.cross_entropy_loss', 'mmdet/models/losses/cross_entropy_loss.py')
    l1 = _load('mmdet.models.losses.smooth_l1_loss', 'mmdet/models/losses/smooth_l1_loss.py')
    acc = _load('mmdet.models.losses.accuracy', 'mmdet/models/losses/accuracy.py')
    return types.SimpleNamespace(
        bbox_overlaps=iou.bbox_overlaps, bbox2delta=coder.bbox2delta, delta2bbox=coder.delta2bbox,
        DeltaXYWHBBoxCoder=coder.DeltaXYWHBBoxCoder, MaxIoUAssigner=asg.MaxIoUAssigner,
        AnchorGenerator=anc.AnchorGenerator, anchor_inside_flags=autil.anchor_inside_flags,
        CrossEntropyLoss=ce.CrossEntropyLoss, L1Loss=l1.L1Loss, SmoothL1Loss=l1.SmoothL1Loss, accuracy=acc.accuracy)


def load_reference_head_train():
    """The training side of the reference's BBoxHead (`_get_target_single`, `get_targets`, `loss`, `regress_by_class`,
    `refine_bboxes`: bbox_head.py:86-352) and its samplers (core/bbox/samplers/{sampling_result, base_sampler,
    random_sampler}.py), for the detector-harness fixtures.  bbox_head.py binds `build_bbox_coder`, `build_loss`,
    `multi_apply` and `accuracy` from packages that are stand-ins here: they are pointed at the reference's own coder / loss
    classes (load_reference_harness); `multi_apply` (core/utils/misc.py, which imports mmcv) is its three-line map / zip."""
    import functools
    hs = load_reference_harness()
    ref = load_reference()
    bh = sys.modules['mmdet.models.roi_heads.bbox_heads.bbox_head']

    def build_bbox_coder(cfg):
        cfg = dict(cfg)
        kind = cfg.pop('type')
        assert kind == 'DeltaXYWHBBoxCoder', kind
        return hs.DeltaXYWHBBoxCoder(**cfg)

    def build_loss(cfg):
        cfg = dict(cfg)
        return {'CrossEntropyLoss': hs.CrossEntropyLoss, 'SmoothL1Loss': hs.SmoothL1Loss, 'L1Loss': hs.L1Loss}[cfg.pop('type')](**cfg)

    def multi_apply(func, *args, **kwargs):
        pfunc = functools.partial(func, **kwargs) if kwargs else func
        return tuple(map(list, zip(*map(pfunc, *args))))

    bh.build_bbox_coder, bh.build_loss, bh.multi_apply, bh.accuracy = build_bbox_coder, build_loss, multi_apply, hs.accuracy
    _mod('mmdet.core.bbox.samplers')
    sys.modules['mmdet.core.bbox'].demodata = _load('mmdet.core.bbox.demodata', 'mmdet/core/bbox/demodata.py')   # RandomSampler's rng helper
    sr = _load('mmdet.core.bbox.samplers.sampling_result', 'mmdet/core/bbox/samplers/sampling_result.py')
    _load('mmdet.core.bbox.samplers.base_sampler', 'mmdet/core/bbox/samplers/base_sampler.py')
    rs = _load('mmdet.core.bbox.samplers.random_sampler', 'mmdet/core/bbox/samplers/random_sampler.py')
    return types.SimpleNamespace(BBoxHead=ref.BBoxHead, RandomSampler=rs.RandomSampler, SamplingResult=sr.SamplingResult,
                                 MaxIoUAssigner=hs.MaxIoUAssigner)


def load_reference_roi_align():
    """The compiled reference RoIAlign (oracle/_ref, built by oracle/build_ref.py)."""
    here = os.path.dirname(os.path.abspath(__file__))
    d = os.path.join(here, '_ref')
    cands = [f for f in os.listdir(d)] if os.path.isdir(d) else []
    so = [f for f in cands if f.startswith('arfe_ref_roi_align') and f.endswith('.so')]
    if not so:
        return None
    spec = importlib.util.spec_from_file_location('arfe_ref_roi_align', os.path.join(d, so[0]))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m
